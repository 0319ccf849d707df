"""Phase split of the on-chip kernel for 2-4 s rows (needs tools/build_variant.sh stamps -DF2_STAMPS). Never used for timing."""
import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from f2cnn_amd import build
build.LIB_PATH = os.path.abspath(os.environ.get("F2CNN_PROBE_LIB", "/root/repo/tools/libf2cnn_hip_stamps.so"))
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
C = 128
ctx = _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
for n in [int(a) for a in sys.argv[1:]] or [65536, 40000]:
    B = 256
    waves = bench.synth_batch(7, 0, B, n)
    off = np.arange(B + 1, dtype=np.int64) * n
    d_wave = ctx.malloc(waves.nbytes); ctx.h2d(d_wave, waves)
    d_out = ctx.malloc(8 * C * n * B)
    print("n =", n, flush=True)
    for _ in range(2):
        ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 0, d_out, None, 1)
    ctx.synchronize()
    ctx.free(d_wave); ctx.free(d_out)
