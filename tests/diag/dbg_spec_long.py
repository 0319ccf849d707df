"""Spot check of the long-row spectral kernel (rows of 32769 .. 65472 samples) against the two-kernel route of the same
library (spectral = 0) and, for one length, against the oracle. Diagnostic, GPU box only."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import f2cnn_oracle as orc
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
ctx = _lib.default_context()
C = 128
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))


def run(ws, lpf, **o):
    offs = np.concatenate([[0], np.cumsum([len(w) for w in ws])]).astype(np.int64)
    wave = np.concatenate(ws)
    env = np.full(C * len(wave), np.nan)
    with ctx.options(**o):
        ctx.filterbank_envelope_fused(wave, _lib.WAVE_I16, offs, coefs, len(ws), C, lpf, 50.0, _lib.FFT_F32, env, None, _lib.MEM_HOST)
        fl = ctx.get_option("spectral_flagged")
    return [env[C * offs[i]:C * offs[i + 1]].reshape(C, -1) for i in range(len(ws))], fl


for lens in ([40000], [32769, 65472], [50001, 33333, 64000, 16000]):
    ws = [orc.synth_utterance(7 + i, n) for i, n in enumerate(lens)]
    for lpf in (False, True):
        g, fl = run(ws, lpf, spectral=1, spectral_min_rows=0)
        r, _ = run(ws, lpf, spectral=0)
        for n, a, b2 in zip(lens, g, r):
            e = np.abs(a - b2).max(axis=1) / np.abs(b2).max(axis=1)
            print(f"n={n} lpf={lpf} flagged={fl} max rel diff vs two-kernel route {e.max():.3e} (chan {e.argmax()}) nan {np.isnan(a).sum()}", flush=True)
x = orc.synth_utterance(3, 40000)
ref = orc.filter_and_envelope(x, coefs, True, 50)
g, fl = run([x], True, spectral=1, spectral_min_rows=0)
e = np.abs(g[0] - ref).max(axis=1) / np.abs(ref).max(axis=1)
print("oracle, n=40000 lpf: max rel err", e.max(), "flagged", fl)
