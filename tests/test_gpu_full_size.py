"""BASELINE-sized runs on the GPU (device buffers through the C ABI), checked through size-independent properties
and spot rows against the oracle: cfg2 (256 x 1 s, 128 channels, filterbank) and a 256-utterance slice of cfg3."""
import numpy as np
import pytest

import bench
import f2cnn_oracle as orc
from conftest import chan_relerr
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters

pytestmark = pytest.mark.gpu
B, C, N = 256, 128, 16000


@pytest.fixture(scope="module")
def corpus():
    ctx = _lib.default_context()
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    waves = bench.synth_batch(2026, 0, B, N)
    off = np.arange(B + 1, dtype=np.int64) * N
    d_wave = ctx.malloc(waves.nbytes)
    ctx.h2d(d_wave, waves)
    yield ctx, coefs, waves, off, d_wave
    ctx.free(d_wave)


def fetch_rows(ctx, dptr, b, rows):
    out = np.empty((len(rows), N))
    for i, c in enumerate(rows):
        ctx.d2h(out[i], dptr + 8 * ((b * C + c) * N))
    return out


def test_cfg2_filterbank_full_batch(corpus):
    ctx, coefs, waves, off, d_wave = corpus
    d_gfb = ctx.malloc(8 * B * C * N)
    ctx.memset(d_gfb, 0xFF, 8 * B * C * N)          # NaN pattern: every element must be overwritten
    ctx.erb_filterbank_batch(d_wave, _lib.WAVE_I16, off, coefs, B, C, d_gfb, _lib.MEM_DEVICE)
    ctx.synchronize()
    rows = [0, 1, 63, 64, 126, 127]
    for b in (0, 100, 255):
        got = fetch_rows(ctx, d_gfb, b, rows)
        assert chan_relerr(got, orc.erb_filterbank(waves[b], coefs[rows])) <= 1e-9
    # batching must not matter: utterance 255 alone gives the same bits as inside the batch when it runs the same
    # (serial) kernel, and the same values to float64 rounding through the time-split path a single utterance takes
    with ctx.options(k1_split=0):
        alone = filters.erb_filterbank(waves[255], coefs)
    np.testing.assert_array_equal(alone[rows], fetch_rows(ctx, d_gfb, 255, rows))
    assert chan_relerr(filters.erb_filterbank(waves[255], coefs)[rows], alone[rows]) <= 1e-11
    # no element left untouched anywhere in the 4.2 GB output (sum of a checksum per utterance is finite)
    whole = np.empty((C, N))
    sums = []
    for b in range(0, B, 17):
        ctx.d2h(whole, d_gfb + 8 * b * C * N)
        assert np.isfinite(whole).all()
        sums.append(float(np.abs(whole).sum()))
    assert np.isfinite(sum(sums)) and min(sums) > 0
    ctx.free(d_gfb)


def test_cfg3_fused_slice_matches_two_step_and_oracle(corpus):
    ctx, coefs, waves, off, d_wave = corpus
    d_env = ctx.malloc(8 * B * C * N)
    d_ref = ctx.malloc(8 * B * C * N)
    ctx.filterbank_envelope_fused(d_wave, _lib.WAVE_I16, off, coefs, B, C, True, 50.0, _lib.FFT_F32, d_env, None,
                                  _lib.MEM_DEVICE)
    # the same through the two public steps with the float64 hand-off
    ctx.erb_filterbank_batch(d_wave, _lib.WAVE_I16, off, coefs, B, C, d_ref, _lib.MEM_DEVICE)
    ctx.envelope_batch(d_ref, off, B, C, True, 50.0, _lib.FFT_F32, d_ref, _lib.MEM_DEVICE)
    ctx.synchronize()
    rows = [0, 31, 64, 127]
    for b in (0, 77, 255):
        fused = fetch_rows(ctx, d_env, b, rows)
        two = fetch_rows(ctx, d_ref, b, rows)
        ref = orc.extract_envelope_from_matrix(orc.erb_filterbank(waves[b], coefs[rows]), True, 50)
        assert chan_relerr(fused, ref) <= 1e-5
        assert chan_relerr(two, ref) <= 1e-5
        assert chan_relerr(fused, two) <= 2e-6      # float32 vs float64 hand-off
        assert (fused > 0).all()                    # envelopes of noise are strictly positive (normalizeInput needs it)
    ctx.free(d_env)
    ctx.free(d_ref)


def test_cfg3_full_batch_of_1000():
    """The benchmark's own workload (bench.py default: 1000 x 1 s, seed 2027, 128 channels, LPF 50, float32 FFT) through
    the fused call: spot rows against the oracle, every output element written, finite and positive."""
    ctx = _lib.default_context()
    Bf = 1000
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    waves = bench.synth_batch(2027, 0, Bf, N)
    off = np.arange(Bf + 1, dtype=np.int64) * N
    d_wave, d_env = ctx.malloc(waves.nbytes), ctx.malloc(8 * Bf * C * N)
    ctx.h2d(d_wave, waves)
    ctx.memset(d_env, 0xFF, 8 * Bf * C * N)
    ctx.filterbank_envelope_fused(d_wave, _lib.WAVE_I16, off, coefs, Bf, C, True, 50.0, _lib.FFT_F32, d_env, None,
                                  _lib.MEM_DEVICE)
    ctx.synchronize()
    rows = [0, 64, 126, 127]
    for b in (0, 499, 999):
        ref = orc.extract_envelope_from_matrix(orc.erb_filterbank(waves[b], coefs[rows]), True, 50)
        assert chan_relerr(fetch_rows(ctx, d_env, b, rows), ref) <= 1e-5
    whole = np.empty((C, N))
    total = 0.0
    for b in range(0, Bf, 41):
        ctx.d2h(whole, d_env + 8 * b * C * N)
        assert np.isfinite(whole).all() and (whole > 0).all()
        total += float(whole.sum())
    assert np.isfinite(total)
    ctx.free(d_wave)
    ctx.free(d_env)


@pytest.mark.parametrize("workload", ["cfg5", "cfg5r"])
def test_cfg5_corpus_sharded_over_two_ranks(workload):
    """BASELINE config 5 in miniature, through bench.py's own job object: the corpus (utterance u = seed 2029 + u;
    cfg5r: U[16000, 64000] samples) sharded r::2, every utterance processed exactly once, envelopes equal to the
    oracle and bit-identical to what a single rank computes for the same utterance (sharding and batching do not
    change results)."""
    ctx = _lib.default_context()
    corpus, Cc = 12, 128
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, Cc, 100))
    all_lens = bench.ragged_lengths(2029, corpus) if workload == "cfg5r" else np.full(corpus, N, np.int64)

    def run(world):
        out = {}
        for rank in range(world):
            idx = np.arange(corpus)[rank::world]
            lens = all_lens[idx]
            waves = bench.synth_corpus(2029, idx, lens, 1)
            job = bench.DspJob(ctx, coefs, Cc, waves, lens, 5, "both", 50, _lib.FFT_F32)   # launches of <= 5 utterances
            job.step()
            ctx.synchronize()
            assert sum(nb for _, _, nb in job.batches) == len(idx)
            pos = 0
            for (w0, off, nb), dst in zip(job.batches, job.out_ptr):
                for k in range(nb):
                    u = int(idx[pos + k])
                    n = int(off[k + 1] - off[k])
                    env = np.empty((Cc, n))
                    ctx.d2h(env, dst + 8 * Cc * int(off[k]))
                    assert u not in out
                    out[u] = env
                pos += nb
            job.free()
        return out

    one, two = run(1), run(2)
    assert sorted(one) == sorted(two) == list(range(corpus))
    for u in range(corpus):
        np.testing.assert_array_equal(one[u], two[u])
    rows = [0, 63, 127]
    for u in (0, 5, 11):
        wave = bench.synth_utterance(2029 + u, int(all_lens[u]))
        ref = orc.extract_envelope_from_matrix(orc.erb_filterbank(wave, coefs[rows]), True, 50)
        assert chan_relerr(one[u][rows], ref) <= 1e-5, (workload, u)


def test_maximum_row_length():
    """The longest rows the library takes (2^22 samples = 4.4 min of audio; limit of the global-memory transform) and the
    first length beyond: a 128-channel filterbank of such an utterance is 4.3 GB, past the 32-bit byte offsets the store
    phase of K1 normally uses (its 64-bit path), and the envelope runs the radix-16 passes over global memory."""
    from f2cnn_amd.scripts.processing.EnvelopeExtraction import ExtractEnvelopeFromMatrix
    nmax = 1 << 22
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
    wave = bench.synth_utterance(4242, nmax)
    gfb = filters.erb_filterbank(wave, coefs)
    assert gfb.shape == (C, nmax)
    rows = [0, 63, 64, 127]
    ref = orc.erb_filterbank(wave, coefs[rows])
    assert chan_relerr(gfb[rows], ref) <= 1e-9
    assert np.isfinite(gfb[:, ::4099]).all() and np.abs(gfb[:, -4096:]).max(axis=1).min() > 0   # every row written to its end
    for lpf in (False, True):
        env = ExtractEnvelopeFromMatrix(gfb[rows], lpf, 50)
        assert chan_relerr(env, orc.extract_envelope_from_matrix(ref, lpf, 50)) <= 1e-5
    del gfb
    with pytest.raises(_lib.F2Error, match="not supported"):
        ExtractEnvelopeFromMatrix(np.ones((1, nmax + 1)), False)
