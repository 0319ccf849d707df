"""The oracle (oracle/f2cnn_oracle.py) against the golden vectors produced by the reference itself
(tests/golden/make_golden.py). CPU only."""
import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import chan_relerr

CASES = ["n1000_c8", "n4096_c8", "n4097_c8", "n16000_c128", "n2500_c128", "impulse_c8", "sine1k_c8",
         "n1500_f64_c8"]


@pytest.mark.parametrize("C", [8, 64, 128])
def test_filter_design_matches_reference(golden, C):
    cf = orc.centre_freqs(16000, C, 100)
    np.testing.assert_allclose(cf, golden[f"g1_cf_{C}"], rtol=1e-14, atol=0)
    co = orc.make_erb_filters(16000, cf)
    assert co.shape == (C, 10)
    np.testing.assert_allclose(co, golden[f"g1_coefs_{C}"], rtol=1e-13, atol=0)


def test_kat_centre_freqs():
    # SURVEY section 4, KAT 1
    for C, first in ((64, 7596.247741), (128, 7795.584900)):
        cf = orc.centre_freqs(16000, C, 100)
        assert np.all(np.diff(cf) < 0)
        assert abs(cf[-1] - 100.0) < 1e-9
        assert abs(cf[0] - first) < 1e-5


def test_kat_unit_gain_at_centre_frequency():
    # KAT 2: |prod_k H_k(e^{j w_cf})| / gain == 1 for every channel
    for C in (64, 128):
        cf = orc.centre_freqs(16000, C, 100)
        co = orc.make_erb_filters(16000, cf)
        z = np.exp(-1j * 2 * np.pi * cf / 16000)
        H = np.ones(C, complex)
        for k in range(1, 5):
            H *= (co[:, 0] + co[:, k] * z + co[:, 5] * z * z) / (co[:, 6] + co[:, 7] * z + co[:, 8] * z * z)
        np.testing.assert_allclose(np.abs(H) / co[:, 9], 1.0, rtol=1e-9)


def test_kat_butter_closed_form():
    for fc in (50, 100, 400):
        b, a = orc.butter1(fc)
        b2, a2 = orc.butter1_closed_form(fc)
        np.testing.assert_allclose(b, b2, rtol=1e-13)
        np.testing.assert_allclose(a, a2, rtol=1e-13)
        assert abs(b.sum() / a.sum() - 1) < 1e-13
    b, a = orc.butter1(50)
    assert abs(b[0] - 0.00972234) < 1e-8 and abs(a[1] + 0.98055532) < 1e-8


@pytest.mark.parametrize("tag", CASES)
def test_filterbank_matches_reference(golden, tag):
    C = 128 if "c128" in tag else 8
    coefs = golden[f"g1_coefs_{C}"]
    rows = golden[f"g2_{tag}_rows"]
    got = orc.erb_filterbank(golden[f"g2_{tag}_wave"], coefs[rows])
    ref = golden[f"g2_{tag}_gfb"]
    assert got.shape == ref.shape and got.dtype == np.float64
    assert chan_relerr(got, ref) <= 1e-13


def test_filterbank_explicit_recurrence(golden):
    # KAT 4: DF2T recurrences == lfilter cascade
    coefs = golden["g1_coefs_8"]
    wave = golden["g2_n1000_c8_wave"]
    assert chan_relerr(orc.erb_filterbank_loops(wave, coefs), golden["g2_n1000_c8_gfb"]) <= 5e-13


@pytest.mark.parametrize("tag", CASES)
def test_envelope_matches_reference(golden, tag):
    gfb = golden[f"g2_{tag}_gfb"]
    for lpf in (0, 50, 100):
        key = f"g3_{tag}_env_lpf{lpf}"
        if key not in golden:
            continue
        got = orc.extract_envelope_from_matrix(gfb, bool(lpf), lpf or 100)
        assert chan_relerr(got, golden[key]) <= 1e-13


def test_kat_padded_hilbert_definition():
    # KAT 5: ifft(fft(pad(x,M)) * h)[:N]
    for n, M in ((16000, 16384), (16384, 16384), (16385, 32768), (1000, 1024)):
        assert orc.padded_length(n) == M
    x = np.random.default_rng(3).standard_normal(1000)
    M = 1024
    h = np.zeros(M)
    h[0] = h[M // 2] = 1
    h[1:M // 2] = 2
    z = np.fft.ifft(np.fft.fft(np.hstack((x, np.zeros(M - 1000)))) * h)[:1000]
    np.testing.assert_allclose(orc.padded_hilbert(x), z, atol=2e-12)


def test_normalize_input_matches_reference(golden):
    np.testing.assert_array_equal(orc.normalize_input(golden["g6_in_f64"]), golden["g6_out_f64"])
    got32 = orc.normalize_input(golden["g6_in_f32"])
    assert got32.dtype == np.float32
    np.testing.assert_array_equal(got32, golden["g6_out_f32"])
    np.testing.assert_array_equal(orc.normalize_input(np.full((11, 128), 3.25)), golden["g6_const_out"])
    assert int(golden["g6_nonpositive_raises"]) == 1
    with pytest.raises(ValueError):
        orc.normalize_input(np.zeros((11, 128)))


def g4_inputs(golden):
    """Rebuild the crafted corpus of make_golden.py: {file key: env}, ordered (key, centre) entries."""
    envs = {k: np.random.default_rng(int(s)).random((128, 16000)) + 0.01
            for k, s in zip(golden["g4_env_keys"], golden["g4_env_seeds"])}
    per_file = {}
    for line in golden["g4_csv_rows"]:
        s, r, sp, se, _ph, tp = str(line).split(",")
        per_file.setdefault(f"{s}/{r}.{sp}.{se}", []).append(int(tp))
    return envs, per_file


def test_gather_windows_matches_reference(golden):
    # KAT 7: files in sorted order, CSV order within a file, float32 cast
    envs, per_file = g4_inputs(golden)
    blocks = [orc.gather_windows(envs[k], per_file[k]) for k in sorted(per_file)]
    got = np.concatenate(blocks).astype(np.float32)
    ref = golden["g4_input_data"]
    assert got.shape == ref.shape == (12, 11, 128)
    np.testing.assert_array_equal(got, ref)
    assert int(golden["g4_last_input_equal"]) == 1


@pytest.mark.parametrize("tag,lpf", [("nolpf", False), ("lpf50", True)])
def test_eval_tensor_matches_the_reference_predict_argument(golden_eval, tag, lpf):
    """G5: what the reference's unmodified EvaluateOneWavFile hands to model.predict (Evaluating.py:42-86) for one 1 s RIFF
    file - filterbank, envelope, every-sample gather, per-window normalizeInput, reshape - against the oracle chain from
    the same samples. Four whole windows and the sum of every one of the 14 240 windows."""
    wave = golden_eval["g5_wave"]
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100))
    x = orc.eval_input_tensor(orc.filter_and_envelope(wave, coefs, lpf, 50))
    assert x.dtype == np.float64 and list(x.shape) == list(golden_eval[f"g5_{tag}_shape"])
    nb = x.shape[0]
    np.testing.assert_allclose(x[[0, 1, 7000, nb - 1]], golden_eval[f"g5_{tag}_windows"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(x.reshape(nb, -1).sum(axis=1), golden_eval[f"g5_{tag}_window_sums"], rtol=1e-12)


def test_eval_tensor_shape_and_range():
    # KAT 6 (restated; the reference's Evaluating.py cannot be imported here, see make_golden.py)
    env = np.random.default_rng(0).random((16, 2000)) + 0.5
    x = orc.eval_input_tensor(env)
    assert x.shape == (2000 - 1760, 11, 16, 1)
    assert np.allclose(x.min(axis=(1, 2, 3)), 0) and np.allclose(x.max(axis=(1, 2, 3)), 1)
    i, k, c = 17, 3, 5
    w = env[:, 800 + i + (np.arange(11) - 5) * 160].T
    np.testing.assert_allclose(x[i, k, c, 0], orc.normalize_input(w)[k, c])


def test_cnn_forward_shapes_and_softmax():
    w = orc.glorot_weights(7)
    assert sum(v.size for v in w.values()) == 1057262        # SURVEY a13 parameter count
    x = np.random.default_rng(1).random((5, 11, 128, 1)).astype(np.float32)
    s = orc.cnn_forward(x, w)
    assert s.shape == (5, 2) and s.dtype == np.float32
    np.testing.assert_allclose(s.sum(axis=1), 1, rtol=1e-6)
    assert orc.labels_from_scores(np.array([[0.5, 0.5], [0.2, 0.8], [0.9, 0.1]])).tolist() == [0, 1, 0]


@pytest.mark.parametrize("rows,channels", [(11, 128), (13, 40), (10, 100)])
def test_cnn_forward_against_an_independent_operator_library(rows, channels):
    """No Keras exists here to pin the CNN restatement, so it is at least held against a second,
    independent implementation of the operators Training.py:93-114 names: torch.nn.functional's conv2d ('same' = one zero
    pixel, 'valid' = none), max_pool2d(2) (floor), channels-last Flatten, linear, softmax - float32 on the CPU."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    w = orc.glorot_weights(3, nchannels=channels, rows=rows, zero_bias=False)
    x = np.random.default_rng(8).random((7, rows, channels)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    conv = lambda h, name, pad: F.conv2d(h, t(w[name + "_w"].transpose(3, 2, 0, 1)), t(w[name + "_b"]), padding=pad)   # HWIO -> OIHW
    with torch.no_grad():
        h = t(x)[:, None]                                        # NCHW, one input channel
        h = F.relu(conv(h, "conv1", 1))
        h = F.max_pool2d(F.relu(conv(h, "conv2", 0)), 2)
        h = F.relu(conv(h, "conv3", 1))
        h = F.max_pool2d(F.relu(conv(h, "conv4", 0)), 2)
        h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)        # Keras flattens (H, W, C)
        h = F.relu(h @ t(w["dense1_w"]) + t(w["dense1_b"]))
        s = F.softmax(h @ t(w["dense2_w"]) + t(w["dense2_b"]), dim=1).numpy()
    np.testing.assert_allclose(orc.cnn_forward(x, w), s, atol=2e-6)
