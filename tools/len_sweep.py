"""Throughput of the fused filterbank + envelope call against utterance length (same total audio per batch).
Diagnostic: shows where rows leave the LDS-resident kernel (> 32768 samples) for the global-memory path."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
import os
if os.environ.get("F2CNN_PROBE_LIB"):
    from f2cnn_amd import build
    build.LIB_PATH = os.path.abspath(os.environ["F2CNN_PROBE_LIB"])
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench

C = 128
TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000      # samples per batch
ctx = _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
LENGTHS = [int(a) for a in sys.argv[2:]] or [8000, 16000, 16384, 24000, 32768, 40000, 48000, 65536, 80000, 131072]
for n in LENGTHS:
    B = max(1, TOTAL // n)
    waves = bench.synth_batch(7, 0, B, n)
    off = np.arange(B + 1, dtype=np.int64) * n
    d_wave = ctx.malloc(waves.nbytes); ctx.h2d(d_wave, waves)
    d_out = ctx.malloc(8 * C * n * B)
    run = lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 0, d_out, None, 1)
    run(); ctx.synchronize(); ctx.prof_enable(True)
    for _ in range(5): run()
    p = ctx.prof_get(); ctx.prof_enable(False)
    ms = {k: t / 5 for k, (c, t) in p.items()}     # per call (a kernel id may cover several launches per call)
    tot = sum(ms.values())
    print(f"n={n:7d} B={B:5d}  " + "  ".join(f"{k} {v:7.3f} ms" for k, v in ms.items()) +
          f"   => {B * n / 16000 / (tot / 1e3):9.0f} audio-s/s (kernel time only)", flush=True)
    ctx.free(d_wave); ctx.free(d_out)
