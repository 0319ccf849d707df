import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
ctx = _lib.Context(0)
for C, B in ((64, 1), (128, 1), (128, 8)):
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    wave = bench.synth_batch(1234, 0, B, 16000).reshape(-1)
    off = np.arange(B + 1, dtype=np.int64) * 16000
    d_wave = ctx.malloc(wave.nbytes); ctx.h2d(d_wave, wave)
    d_out = ctx.malloc(8 * C * 16000 * B)
    for K in (-1, 16, 32, 48, 64):
        ctx.set_option("k1_split", K)
        run = lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 0, d_out, None, 1)
        for _ in range(20):
            run(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            run(); ctx.synchronize()
        dt = (time.perf_counter() - t0) / 300
        ctx.prof_enable(True)
        for _ in range(50):
            run()
        p = ctx.prof_get(); ctx.prof_enable(False)
        print(f"C={C} B={B} k1_split={K}: {dt * 1e6:.1f} us per call;", {k: round(t / 50 * 1e3, 1) for k, (c, t) in p.items()}, flush=True)
    ctx.free(d_wave); ctx.free(d_out)
