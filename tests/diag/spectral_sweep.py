"""Randomised agreement sweep: one-kernel spectral route against the two-kernel route of the same library over random
channel counts, batch shapes, lengths (all four length classes, class edges), wave types, low-pass settings. Diagnostic."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters

ctx = _lib.default_context()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
EDGES = [4097, 8192, 8193, 16384 - 64, 16385, 32768 - 64, 32769, 65472, 40000, 50001, 61441, 61439]
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    C = int(rng.choice([1, 3, 8, 31, 64, 70, 128, 130]))
    B = int(rng.integers(1, 5))
    lens = [int(rng.choice(EDGES)) if rng.random() < 0.5 else int(rng.integers(4097, 65473)) for _ in range(B)]
    f64 = rng.random() < 0.3
    lpf = rng.random() < 0.7
    cutoff = float(rng.choice([50.0, 100.0, 20.0]))
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, int(rng.choice([50, 100, 300]))))
    waves = [np.clip(np.round(rng.standard_normal(n) * 3000), -32768, 32767) for n in lens]
    if rng.random() < 0.2:
        waves[0][: lens[0] // 2] = 0                       # a silent half
    flat = np.concatenate(waves).astype(np.float64 if f64 else np.int16)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    res = []
    for sp in (1, 0):
        env = np.full(C * int(offs[-1]), np.nan)
        with ctx.options(spectral=sp, spectral_min_rows=0):
            ctx.filterbank_envelope_fused(flat, _lib.WAVE_F64 if f64 else _lib.WAVE_I16, offs, coefs, B, C, lpf, cutoff, _lib.FFT_F32,
                                          env, None, _lib.MEM_HOST)
            fl = ctx.get_option("spectral_flagged") if sp else 0
        res.append(env)
    err = 0.0
    for b in range(B):
        a = res[0][C * offs[b]:C * offs[b + 1]].reshape(C, -1)
        r = res[1][C * offs[b]:C * offs[b + 1]].reshape(C, -1)
        err = max(err, float((np.abs(a - r).max(axis=1) / np.abs(r).max(axis=1)).max()))
    worst = max(worst, err)
    flag = "" if err <= 4e-6 and not np.isnan(res[0]).any() else "   <-- CHECK"
    print(f"C={C:3d} lens={lens} f64={int(f64)} lpf={int(lpf)} cutoff={cutoff:g} flagged={int(fl)} max rel diff {err:.2e}{flag}", flush=True)
print("worst", worst)
