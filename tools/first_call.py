"""Time of the first fused call of a context per length class (table construction) against a repeated one. Diagnostic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
C = 128
ctx = _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
for n in (16000, 30000, 60000):
    B = 40
    wave = bench.synth_batch(7, 0, B, n)
    off = np.arange(B + 1, dtype=np.int64) * n
    d_wave = ctx.malloc(wave.nbytes); ctx.h2d(d_wave, wave)
    d_out = ctx.malloc(8 * C * n * B)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 0, d_out, None, 1); ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"n = {n}: first call {ts[0] * 1e3:.1f} ms, then {ts[1] * 1e3:.2f}, {ts[2] * 1e3:.2f} ms", flush=True)
    ctx.free(d_wave); ctx.free(d_out)
