"""Seeded random ragged batches through the fused C-ABI call against the oracle: lengths around every size class of K2
(LDS-resident plans, the 1024-thread plan, the four-step path), channel counts that are not multiples of 64, with and
without the low-pass, both FFT precisions, int16 and float64 waves, with and without the GFB output."""
import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import chan_relerr
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters

pytestmark = pytest.mark.gpu

LENGTH_POOL = [1, 2, 3, 5, 17, 63, 64, 65, 255, 257, 1000, 1023, 1025, 2049, 4097, 8191, 8193, 9999, 16383, 16384, 16385,
               20001, 32767, 32768, 32769, 40001, 65535, 65537, 70000]


@pytest.mark.parametrize("seed", range(40))
def test_fused_random_ragged_batches(seed):
    """(Half of the seeds let every eligible row - 4097..32768 samples, make_erb_filters table, float FFT, no GFB output -
    take the one-kernel spectral route whatever the batch size - seeds 6, 9, 18, 21, 30, 33 have such rows, with and without
    the low-pass; the others leave these small batches to the two-kernel route.)"""
    rng = np.random.default_rng(1000 + seed)
    ctx = _lib.default_context()
    ctx.set_option("spectral_min_rows", 0 if seed % 4 in (1, 2) else 4096)
    Cn = int(rng.choice([1, 3, 7, 20, 64, 65, 70]))
    B = int(rng.integers(1, 7))
    lens = [int(rng.choice(LENGTH_POOL)) for _ in range(B)]
    if seed % 4 == 0:
        lens[rng.integers(0, B)] = 0                      # an empty utterance inside the batch
    lpf = bool(seed % 2)
    cutoff = float(rng.choice([20, 50, 100, 400]))
    precision = _lib.FFT_F64 if seed % 5 == 4 else _lib.FFT_F32
    as_f64 = seed % 3 == 2
    want_gfb = seed % 3 != 0
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, Cn, 100))
    waves = [orc.synth_utterance(5000 + 10 * seed + i, n) for i, n in enumerate(lens)]
    flat = np.concatenate(waves) if sum(lens) else np.zeros(0, np.int16)
    if as_f64:
        flat = flat.astype(np.float64) * 0.37
        waves = [w.astype(np.float64) * 0.37 for w in waves]
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    env = np.full(Cn * int(off[-1]), np.nan)
    gfb = np.full(Cn * int(off[-1]), np.nan) if want_gfb else None
    ctx.filterbank_envelope_fused(flat, _lib.WAVE_F64 if as_f64 else _lib.WAVE_I16, off, coefs, B, Cn, lpf, cutoff if lpf else 0.0,
                                  precision, env, gfb, _lib.MEM_HOST)
    tol_env = 1e-5 if precision == _lib.FFT_F32 else 1e-10
    for b, w in enumerate(waves):
        n = lens[b]
        if n == 0:
            continue
        ref_gfb = orc.erb_filterbank(w, coefs)
        ref_env = orc.extract_envelope_from_matrix(ref_gfb, lpf, cutoff)
        got_env = env[Cn * off[b]:Cn * off[b + 1]].reshape(Cn, n)
        assert not np.isnan(got_env).any(), (seed, b, n)
        assert chan_relerr(got_env, ref_env) <= tol_env, (seed, b, n, Cn, lpf, cutoff, precision)
        if want_gfb:
            got_gfb = gfb[Cn * off[b]:Cn * off[b + 1]].reshape(Cn, n)
            assert chan_relerr(got_gfb, ref_gfb) <= 1e-9, (seed, b, n)
    ctx.set_option("spectral_min_rows", 4096)
