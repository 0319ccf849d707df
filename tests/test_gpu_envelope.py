"""K2 parity on the GPU, through the C ABI: Hilbert envelope (+LPF) vs the oracle and golden vectors."""
import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import chan_relerr
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
from f2cnn_amd.scripts.processing import EnvelopeExtraction as EE

pytestmark = pytest.mark.gpu
TOL = 1e-5            # north_star contract, per-channel max-norm
TOL_F64 = 1e-11       # float64 FFT mode

CASES = ["n1000_c8", "n4096_c8", "n4097_c8", "n16000_c128", "n2500_c128", "impulse_c8", "sine1k_c8",
         "n1500_f64_c8"]


@pytest.mark.parametrize("precision,tol", [(_lib.FFT_F32, TOL), (_lib.FFT_F64, TOL_F64)])
@pytest.mark.parametrize("tag", CASES)
def test_golden(golden, tag, precision, tol):
    gfb = golden[f"g2_{tag}_gfb"]
    for lpf in (0, 50, 100):
        key = f"g3_{tag}_env_lpf{lpf}"
        if key not in golden:
            continue
        got = EE.ExtractEnvelopeFromMatrix(gfb, bool(lpf), lpf or 100, precision=precision)
        assert got.shape == gfb.shape and got.dtype == np.float64
        err = chan_relerr(got, golden[key])
        assert err <= tol, (tag, lpf, err)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 64, 65, 100, 128, 255, 256, 257, 511, 513, 1024,
                               1500, 2048, 2049, 4096, 5000, 8192, 8193, 16001, 16384])
def test_all_fft_sizes(N):
    rng = np.random.default_rng(N)
    m = rng.standard_normal((3, N)) * np.array([[1.0], [1e-6], [3000.0]])
    for lpf in (False, True):
        ref = orc.extract_envelope_from_matrix(m, lpf, 50)
        assert chan_relerr(EE.ExtractEnvelopeFromMatrix(m, lpf, 50, precision=_lib.FFT_F64), ref) <= TOL_F64
        assert chan_relerr(EE.ExtractEnvelopeFromMatrix(m, lpf, 50, precision=_lib.FFT_F32), ref) <= TOL


@pytest.mark.parametrize("N", [16385, 20000, 30000, 32767, 32768])
def test_fft32_longest_rows(N):
    m = np.random.default_rng(5).standard_normal((2, N))
    for lpf in (False, True):
        ref = orc.extract_envelope_from_matrix(m, lpf, 50)
        assert chan_relerr(EE.ExtractEnvelopeFromMatrix(m, lpf, 50, precision=_lib.FFT_F32), ref) <= TOL


@pytest.mark.parametrize("N,precision,tol", [(32769, _lib.FFT_F32, TOL), (40000, _lib.FFT_F32, TOL), (48000, _lib.FFT_F32, TOL),
                                             (65536, _lib.FFT_F32, TOL), (70001, _lib.FFT_F32, TOL),
                                             (131073, _lib.FFT_F32, TOL), (262144, _lib.FFT_F32, TOL),
                                             (262145, _lib.FFT_F32, TOL), (20000, _lib.FFT_F64, TOL_F64),
                                             (50000, _lib.FFT_F64, TOL_F64)])
def test_long_rows_beyond_lds(N, precision, tol):
    # TIMIT sentences are 1-8 s. float transforms of 32769..262144 samples run as four-step FFTs (H1 = 8, 16, 32 times
    # LDS-resident 4096-point parts); longer rows and float64 transforms take the global-memory radix-16 passes
    m = np.random.default_rng(N).standard_normal((3, N)) * np.array([[1.0], [3000.0], [1e-3]])
    for lpf in (False, True):
        ref = orc.extract_envelope_from_matrix(m, lpf, 50)
        got = EE.ExtractEnvelopeFromMatrix(m, lpf, 50, precision=precision)
        assert chan_relerr(got, ref) <= tol, (N, lpf)


def test_mixed_short_and_long_batch():
    rng = np.random.default_rng(77)
    mats = [rng.standard_normal((4, n)) for n in (16000, 50000, 1000, 33000, 70000, 49999, 140001)]
    for lpf, cutoff in ((True, 50), (False, 50), (True, 5)):      # 5 Hz: the filter state decays over many segments
        outs = EE.ExtractEnvelopesFromMatrices(mats, lpf, cutoff)
        for m, o in zip(mats, outs):
            assert chan_relerr(o, orc.extract_envelope_from_matrix(m, lpf, cutoff)) <= TOL, (m.shape, lpf, cutoff)


def test_on_filterbank_output_cfg1():
    # BASELINE config 1: 1 s, 64 channels, LPF 50
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, 64, 100))
    wave = orc.synth_utterance(1234, 16000)
    gfb = filters.erb_filterbank(wave, coefs)
    ref = orc.extract_envelope_from_matrix(orc.erb_filterbank(wave, coefs), True, 50)
    assert chan_relerr(EE.ExtractEnvelopeFromMatrix(gfb, True, 50), ref) <= TOL
    assert chan_relerr(EE.ExtractEnvelopeFromMatrix(gfb, False), orc.extract_envelope_from_matrix(gfb)) <= TOL


def test_ragged_batch():
    rng = np.random.default_rng(8)
    mats = [rng.standard_normal((5, n)) for n in (1000, 16000, 3, 4097, 1000)]
    outs = EE.ExtractEnvelopesFromMatrices(mats, True, 100)
    for m, o in zip(mats, outs):
        assert chan_relerr(o, orc.extract_envelope_from_matrix(m, True, 100)) <= TOL


def test_in_place_on_device_and_input_untouched():
    ctx = _lib.default_context()
    m = np.random.default_rng(9).standard_normal((4, 5000))
    keep = m.copy()
    ref = orc.extract_envelope_from_matrix(m, True, 50)
    d = ctx.malloc(m.nbytes)
    ctx.h2d(d, m)
    off = np.array([0, 5000], np.int64)
    ctx.envelope_batch(d, off, 1, 4, True, 50.0, _lib.FFT_F32, d, _lib.MEM_DEVICE)   # env aliases gfb
    out = np.empty_like(m)
    ctx.d2h(out, d)
    ctx.free(d)
    assert chan_relerr(out, ref) <= TOL
    EE.ExtractEnvelopeFromMatrix(m, True, 50)
    assert np.array_equal(m, keep)


def test_properties_full_size():
    # scaling: env(a*x) = |a| env(x); LPF DC gain 1: constant envelope in, same constant out (late samples)
    rng = np.random.default_rng(10)
    m = rng.standard_normal((128, 16000))
    e1 = EE.ExtractEnvelopeFromMatrix(m, True, 50)
    e2 = EE.ExtractEnvelopeFromMatrix(-2.5 * m, True, 50)
    assert chan_relerr(e2, 2.5 * e1) <= 2e-6
    assert np.all(EE.ExtractEnvelopeFromMatrix(m, False) >= np.abs(m) * (1 - 1e-6))


def test_bad_arguments():
    with pytest.raises(_lib.F2Error):
        EE.ExtractEnvelopeFromMatrix(np.ones((2, 10)), True, 9000)
    with pytest.raises(ValueError):
        EE.ExtractEnvelopeFromMatrix(np.ones(10))


@pytest.mark.parametrize("n", [32769, 40000, 48001, 65535, 65536])
def test_rows_of_32769_to_65536_samples_on_chip_and_four_step(n):
    """Rows of 2-4 s stay on the chip (two LDS-resident sub-rows per workgroup, f2_envelope_pair.hip) unless the call is
    in place on float64 rows, which keeps the four-step path: both against the oracle, and against each other."""
    import os
    ctx = _lib.default_context()
    C = 5
    m = np.random.default_rng(n).standard_normal((C, n)) * np.array([[1.0], [3000.0], [1e-3], [40.0], [7.0]])
    off = np.array([0, n], dtype=np.int64)
    d_in, d_out = ctx.malloc(m.nbytes), ctx.malloc(m.nbytes)
    for lpf, cutoff in ((False, 50.0), (True, 50.0), (True, 5.0)):
        ref = orc.extract_envelope_from_matrix(m, lpf, cutoff)
        got = np.empty_like(m)
        ctx.h2d(d_in, m)
        ctx.envelope_batch(d_in, off, 1, C, lpf, cutoff, _lib.FFT_F32, d_out, _lib.MEM_DEVICE)     # on chip (float64 rows)
        ctx.d2h(got, d_out)
        assert chan_relerr(got, ref) <= TOL, (n, lpf, cutoff)
        ctx.envelope_batch(d_in, off, 1, C, lpf, cutoff, _lib.FFT_F32, d_in, _lib.MEM_DEVICE)      # in place: four-step
        four = np.empty_like(m)
        ctx.d2h(four, d_in)
        assert chan_relerr(four, ref) <= TOL, (n, lpf, cutoff)
        assert chan_relerr(got, four) <= 4e-6
    ctx.free(d_in)
    ctx.free(d_out)
    # through the fused call (float32 hand-off rows in the scratch buffer), with and without the on-chip kernel
    from f2cnn_amd.gammatone import filters
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, 8, 100))
    waves = [orc.synth_utterance(n + k, n - 2 * k) for k in range(3)]      # even and odd lengths, odd scratch offsets
    offs = np.concatenate([[0], np.cumsum([len(w) for w in waves])]).astype(np.int64)
    flat = np.concatenate(waves)
    env = np.empty(8 * int(offs[-1]))
    outs = {}
    for tag in ("pair", "four"):
        with ctx.options(env_pair=1 if tag == "pair" else 0):
            ctx.filterbank_envelope_fused(flat, _lib.WAVE_I16, offs, coefs, 3, 8, True, 50.0, _lib.FFT_F32, env, None,
                                          _lib.MEM_HOST)
        outs[tag] = [env[8 * offs[b]:8 * offs[b + 1]].reshape(8, -1).copy() for b in range(3)]
    for b, w in enumerate(waves):
        ref = orc.filter_and_envelope(w, coefs, True, 50)
        assert chan_relerr(outs["pair"][b], ref) <= TOL and chan_relerr(outs["four"][b], ref) <= TOL
        assert chan_relerr(outs["pair"][b], outs["four"][b]) <= 4e-6
