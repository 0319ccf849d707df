"""Filterbank latency at small batch sizes: serial kernel (context option k1_split = 0) against the time-split path."""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
ctx = _lib.Context(0)
for B, C in ((1, 64), (1, 128), (8, 128), (32, 128), (96, 128), (192, 128), (256, 128)):
    N = 16000
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    waves = bench.synth_batch(1234, 0, B, N)
    off = np.arange(B + 1, dtype=np.int64) * N
    d_w = ctx.malloc(waves.nbytes); ctx.h2d(d_w, waves); d_o = ctx.malloc(8 * C * N * B)
    res = {}
    for mode in ("0", ""):
        ctx.set_option("k1_split", int(mode) if mode else -1)
        ctx.set_option("spectral", 0)
        f = lambda: ctx.filterbank_envelope_fused(d_w, 0, off, coefs, B, C, True, 50.0, 0, d_o, None, 1)
        f(); ctx.synchronize(); ctx.prof_enable(True)
        for _ in range(10): f()
        p = ctx.prof_get(); ctx.prof_enable(False)
        res[mode or "split"] = {k: round(ms / n * 1e3, 1) for k, (n, ms) in p.items()}
    print(f"B={B:4d} C={C:4d}  serial {res['0']}  time-split {res['split']}  (us per call)", flush=True)
    ctx.free(d_w); ctx.free(d_o)
