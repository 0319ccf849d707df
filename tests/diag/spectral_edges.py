"""Edge lengths of the long-row spectral kernel (class and bank / plane boundaries, block boundaries of the low-pass sweeps):
spectral route against the two-kernel route of the same library. Diagnostic, GPU box only."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
ctx = _lib.default_context()
rng = np.random.default_rng(4)
worst = 0
for C in (5, 128):
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    for lens in ([32769, 32770, 32771, 32772], [32773, 32774, 32775, 32776], [65465, 65466, 65467, 65468], [65469, 65470, 65471, 65472], [36863, 36864, 36865, 40959, 40960, 40961], [61439, 61440, 61441, 61443]):
        waves = [np.clip(np.round(rng.standard_normal(n) * 3000), -32768, 32767).astype(np.int16) for n in lens]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = np.concatenate(waves)
        for lpf in (True, False):
            res = []
            for sp in (1, 0):
                env = np.full(C * int(offs[-1]), np.nan)
                with ctx.options(spectral=sp, spectral_min_rows=0):
                    ctx.filterbank_envelope_fused(flat, _lib.WAVE_I16, offs, coefs, len(lens), C, lpf, 50.0, _lib.FFT_F32, env, None, _lib.MEM_HOST)
                res.append(env)
            assert not np.isnan(res[0]).any()
            for b in range(len(lens)):
                a = res[0][C * offs[b]:C * offs[b + 1]].reshape(C, -1); r = res[1][C * offs[b]:C * offs[b + 1]].reshape(C, -1)
                e = float((np.abs(a - r).max(axis=1) / np.abs(r).max(axis=1)).max())
                worst = max(worst, e)
                if e > 4e-6: print("CHECK", C, lens[b], lpf, e)
print("edge lengths: worst", worst)
