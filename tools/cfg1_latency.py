"""Single-utterance latency (cfg1: 1 s, 64 channels, fused call + device sync) on the two-kernel route and on the spectral route
(option spectral_min_rows = 0). Diagnostic."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
wave = bench.synth_batch(1234, 0, 1, 16000)
off = np.array([0, 16000], dtype=np.int64)
d_wave = ctx.malloc(wave.nbytes); ctx.h2d(d_wave, wave)
d_out = ctx.malloc(8 * C * 16000)
for rows in (4096, 0):
    ctx.set_option("spectral_min_rows", rows)
    run = lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, 1, C, True, 50.0, 0, d_out, None, 1)
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
    print(f"spectral_min_rows = {rows}: {dt * 1e6:.1f} us per call;", {k: round(t / 50 * 1e3, 1) for k, (c, t) in p.items()}, "us per kernel group", flush=True)
