"""K1 parity on the GPU, through the C ABI: HIP filterbank vs the oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import chan_relerr
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters

pytestmark = pytest.mark.gpu
TOL = 1e-5   # north_star: filterbank outputs within 1e-5 relative (per-channel max-norm)

CASES = ["n1000_c8", "n4096_c8", "n4097_c8", "n16000_c128", "n2500_c128", "impulse_c8", "sine1k_c8",
         "n1500_f64_c8"]


@pytest.mark.parametrize("tag", CASES)
def test_golden(golden, tag):
    C = 128 if "c128" in tag else 8
    coefs = golden[f"g1_coefs_{C}"]
    rows = golden[f"g2_{tag}_rows"]
    got = filters.erb_filterbank(golden[f"g2_{tag}_wave"], coefs[rows])
    ref = golden[f"g2_{tag}_gfb"]
    assert got.shape == ref.shape and got.dtype == np.float64
    err = chan_relerr(got, ref)
    assert err <= TOL, err
    assert err <= 1e-10   # float64 recurrence: far inside the contract


@pytest.mark.parametrize("C,N", [(64, 16000), (128, 16000), (128, 777), (5, 33), (130, 1000), (1, 1), (128, 31)])
def test_vs_oracle(C, N):
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    wave = orc.synth_utterance(1234 + C + N, N)
    got = filters.erb_filterbank(wave, coefs)
    assert chan_relerr(got, orc.erb_filterbank(wave, coefs)) <= 1e-10


def test_ragged_batch_and_empty():
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, 128, 100))
    lens = [1000, 0, 16000, 1, 4097, 63]
    waves = [orc.synth_utterance(50 + i, n) for i, n in enumerate(lens)]
    outs = filters.erb_filterbank_batch(waves, coefs)
    assert [o.shape for o in outs] == [(128, n) for n in lens]
    for w, o in zip(waves, outs):
        if len(w):
            assert chan_relerr(o, orc.erb_filterbank(w, coefs)) <= 1e-10
    assert filters.erb_filterbank_batch([], coefs) == []
    assert filters.erb_filterbank(np.zeros(0, np.int16), coefs).shape == (128, 0)


def test_linearity_and_full_size_properties():
    # size-independent property at a BASELINE-sized row count: filterbank is linear and causal
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, 128, 100))
    a = orc.synth_utterance(1, 16000).astype(np.float64)
    b = orc.synth_utterance(2, 16000).astype(np.float64)
    ya, yb, yab = (filters.erb_filterbank(w, coefs) for w in (a, b, 2 * a - 3 * b))
    assert chan_relerr(yab, 2 * ya - 3 * yb) <= 1e-9
    b2 = b.copy()
    b2[8000:] = 0
    assert np.array_equal(filters.erb_filterbank(b2, coefs)[:, :8000], yb[:, :8000])


def test_bad_arguments():
    ctx = _lib.default_context()
    co = np.zeros((4, 10))
    with pytest.raises(_lib.F2Error):
        ctx.erb_filterbank_batch(np.zeros(4, np.int16), 7, np.array([0, 4], np.int64), co, 1, 4, np.zeros(16), 0)
    with pytest.raises(_lib.F2Error):
        ctx.erb_filterbank_batch(np.zeros(4, np.int16), 0, np.array([1, 4], np.int64), co, 1, 4, np.zeros(16), 0)
    with pytest.raises(ValueError):
        filters.erb_filterbank(np.zeros((2, 2)), co)


def _fb(waves, coefs, split):
    with _lib.default_context().options(k1_split=int(split)):
        return filters.erb_filterbank_batch(waves, coefs)


@pytest.mark.parametrize("C", [64, 128, 8])
def test_time_split_path_for_small_batches(C):
    """B = 1 (cfg1, `cnn eval` of one file) runs the filterbank time-split (segments chained through T^L): same
    results as the serial kernel to float64 rounding, and as the oracle to the parity bar; ragged batches, forced
    segment counts, float64 input."""
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    wave = orc.synth_utterance(1234, 16000)
    serial = _fb([wave], coefs, "0")[0]
    for k in ("", "2", "5", "32"):
        split = _fb([wave], coefs, k)[0] if k else filters.erb_filterbank(wave, coefs)     # "": the default policy
        assert chan_relerr(split, serial) <= 1e-11, k
    assert chan_relerr(filters.erb_filterbank(wave, coefs), orc.erb_filterbank(wave, coefs)) <= 1e-9
    # ragged: lengths around the segment and tile boundaries, one too short to be split, float64 samples
    rng = np.random.default_rng(5)
    waves = [orc.synth_utterance(10 + i, n) for i, n in enumerate((4097, 300, 31, 8192, 5000, 1))]
    for k in ("3", "7", ""):
        got = _fb(waves, coefs, k) if k else filters.erb_filterbank_batch(waves, coefs)
        for w, g in zip(waves, got):
            assert g.shape == (C, len(w))
            assert chan_relerr(g, orc.erb_filterbank(w, coefs)) <= 1e-9
    noisy = [w.astype(np.float64) + rng.standard_normal(len(w)) for w in waves[:4]]
    for g, w in zip(_fb(noisy, coefs, "4"), noisy):
        assert chan_relerr(g, orc.erb_filterbank(w, coefs)) <= 1e-9


def test_queue_mode_for_ragged_batches():
    """Ragged batches hand their (utterance, channel group) units out longest first (option k1_queue = 1 forces the
    path that large ragged batches take by themselves): same bits as one fixed unit per wave."""
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, 128, 100))
    rng = np.random.default_rng(11)
    waves = [orc.synth_utterance(100 + i, int(n)) for i, n in enumerate(rng.integers(40, 3000, size=37))]
    waves.append(np.zeros(0, np.int16))
    ctx = _lib.default_context()
    with ctx.options(k1_split=0):
        plain = filters.erb_filterbank_batch(waves, coefs)
        with ctx.options(k1_queue=1):
            queued = filters.erb_filterbank_batch(waves, coefs)
    for a, b, w in zip(plain, queued, waves):
        assert a.shape == b.shape == (128, len(w))
        np.testing.assert_array_equal(a, b)
    assert chan_relerr(queued[3], orc.erb_filterbank(waves[3], coefs)) <= 1e-9


@pytest.mark.parametrize("C", [128, 130])
@pytest.mark.parametrize("mode", ["plain", "split", "queue"])
def test_rows_that_do_not_start_on_a_line(C, mode):
    """Utterance lengths that are not multiples of 16 (32) put most rows of the (C, n) float64 (float32 hand-off) matrix
    off the 128-byte lines; the kernel then stores every row with its own lag (f2_filterbank.hip). Every element of
    every row against the oracle, for the plain kernel, the time-split path and the unit queue, with the float64
    output, the in-slot float32 hand-off (fused call) and the compact float32 rows long utterances hand over."""
    ctx = _lib.default_context()
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    lens = [1599, 1602, 33, 1601, 47, 3, 2050, 97, 1, 33001 if mode != "split" else 4099]
    waves = [orc.synth_utterance(300 + i, n) for i, n in enumerate(lens)]
    opts = {"plain": dict(k1_split=0, k1_queue=0), "split": dict(k1_split=3), "queue": dict(k1_split=0, k1_queue=1)}[mode]
    # (the fused call below must run the filterbank kernel under test, not the spectral path)
    with ctx.options(spectral=0, **opts):
        gfb = filters.erb_filterbank_batch(waves, coefs)
        # fused call: float32 hand-off (in the rows' own slots up to 32768 samples, compact scratch rows beyond)
        flat = np.concatenate(waves)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        envs = np.full(C * int(off[-1]), np.nan)
        ctx.filterbank_envelope_fused(flat, _lib.WAVE_I16, off, coefs, len(lens), C, False, 0.0, _lib.FFT_F32, envs, None,
                                      _lib.MEM_HOST)
    for i, (w, g) in enumerate(zip(waves, gfb)):
        ref = orc.erb_filterbank(w, coefs)
        assert g.shape == ref.shape
        assert chan_relerr(g, ref) <= 1e-9, (mode, i)
        e = envs[C * off[i]:C * off[i + 1]].reshape(C, len(w))
        assert chan_relerr(e, orc.extract_envelope_from_matrix(ref, False, 0)) <= TOL, (mode, i)


@pytest.mark.parametrize("C,N", [(128, 4001), (70, 16000), (8, 97)])
def test_general_coefficient_tables(C, N):
    """erb_filterbank takes any (C, 10) table, like the reference (filters.py:228-237 hands the columns to lfilter as
    they are): numerators with a z^-2 term and B0 != 1 run the transposed-direct-form-II kernel instead of the
    three-FMA sections make_erb_filters' tables allow; single utterance, a ragged batch and float64 samples."""
    rng = np.random.default_rng(C + N)
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100)).copy()
    coefs[:, 5] = coefs[:, 0] * rng.uniform(-0.5, 0.5, C)          # A2
    scale = rng.uniform(0.5, 2.0, C)                                # B0 (lfilter normalises by it)
    coefs[:, 6:9] *= scale[:, None]
    wave = orc.synth_utterance(77 + C, N)
    ref = orc.erb_filterbank(wave, coefs)
    assert chan_relerr(filters.erb_filterbank(wave, coefs), ref) <= 1e-9
    noisy = wave.astype(np.float64) + rng.standard_normal(N)
    assert chan_relerr(filters.erb_filterbank(noisy, coefs), orc.erb_filterbank(noisy, coefs)) <= 1e-9
    waves = [wave[:n] for n in (N, 33, N // 2 + 1, 1)]
    for w, g in zip(waves, filters.erb_filterbank_batch(waves, coefs)):
        assert chan_relerr(g, orc.erb_filterbank(w, coefs)) <= 1e-9
    if N <= 200:
        assert chan_relerr(orc.erb_filterbank_loops(wave, coefs), ref) <= 1e-12
