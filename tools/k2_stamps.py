"""Run the fused path once with the -DF2_STAMPS diagnostic library (tools/libf2cnn_hip_stamps.so) and print the
mean per-workgroup cycle split of k_envelope. Never used for timing."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from f2cnn_amd import build
import os
build.LIB_PATH = os.path.abspath(os.environ.get("F2CNN_PROBE_LIB", "/root/repo/tools/libf2cnn_hip_stamps.so"))
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
B, C, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 128, int(sys.argv[4]) if len(sys.argv) > 4 else 16000
lpf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.Context(0)
ctx.set_option("spectral", int(sys.argv[3]) if len(sys.argv) > 3 else 1)   # 0: stamps of the filterbank kernel + envelope kernel route
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
waves = bench.synth_batch(2027, 0, B, N)
off = np.arange(B + 1, dtype=np.int64) * N
d_wave = ctx.malloc(waves.nbytes); ctx.h2d(d_wave, waves)
d_b = ctx.malloc(8 * C * N * B)
for _ in range(2):
    ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, bool(lpf), 50.0, 0, d_b, None, 1)
ctx.synchronize()
