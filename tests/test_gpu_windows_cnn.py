"""K3 (window gather + normalise), K4 (CNN forward) and the `cnn eval` pipeline on the GPU, through the C ABI."""
import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import chan_relerr
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
from f2cnn_amd.model import F2CNNModel
from f2cnn_amd.scripts.CNN import Evaluating
from f2cnn_amd.scripts.CNN.Training import normalizeInput
from f2cnn_amd.scripts.processing import InputGenerator
from test_oracle_golden import g4_inputs

pytestmark = pytest.mark.gpu


def oracle_weights(m):
    return dict(m.tensors)


def test_gather_matches_reference_golden(golden):
    envs, per_file = g4_inputs(golden)
    blocks = [InputGenerator.gather_windows(envs[k], per_file[k], 5, 160) for k in sorted(per_file)]
    got = np.concatenate(blocks)
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, golden["g4_input_data"])      # pure gather: bit exact


def test_gather_edges_and_errors():
    env = np.random.default_rng(0).random((7, 2000)) + 0.1
    w = InputGenerator.gather_windows(env, [800, 1199], 5, 160)
    np.testing.assert_array_equal(w, orc.gather_windows(env, [800, 1199]).astype(np.float32))
    for bad in (799, 1200, -5):
        with pytest.raises(_lib.F2Error) as e:
            InputGenerator.gather_windows(env, [bad], 5, 160)
        assert e.value.code == _lib.F2_ERR_INVALID
    assert InputGenerator.gather_windows(env, [], 5, 160).shape == (0, 11, 7)
    # other geometry
    w = InputGenerator.gather_windows(env, [30, 100], 2, 7)
    np.testing.assert_array_equal(w, orc.gather_windows(env, [30, 100], 2, 7).astype(np.float32))


def test_normalize_input_golden(golden):
    got = normalizeInput(golden["g6_in_f64"])
    assert got.dtype == np.float64 and got.shape == (11, 128)
    np.testing.assert_allclose(got, golden["g6_out_f64"], atol=1e-7)     # float32 result of a float64 computation
    got32 = normalizeInput(golden["g6_in_f32"])
    assert got32.dtype == np.float32 and got32.shape == (11, 128, 1)
    np.testing.assert_allclose(got32, golden["g6_out_f32"], atol=3e-7)
    np.testing.assert_array_equal(normalizeInput(np.full((11, 128), 3.25)), golden["g6_const_out"])
    with pytest.raises(ValueError):
        normalizeInput(np.zeros((11, 128)))
    bad = golden["g6_in_f64"].copy()
    bad[3, 5] = -1.0
    with pytest.raises(ValueError):
        normalizeInput(bad)


def test_eval_windows_normalised():
    ctx = _lib.default_context()
    env = np.random.default_rng(2).random((128, 3000)) * 40 + 1e-3
    nb = 3000 - 1760
    out = np.empty((nb, 11, 128), np.float32)
    ctx.gather_windows(env, 128, 3000, None, nb, 5, 160, True, out, _lib.MEM_HOST)
    ref = orc.eval_input_tensor(env)[..., 0]
    np.testing.assert_allclose(out, ref, atol=2e-7)
    assert out.min() == 0.0 and out.max() == 1.0


@pytest.mark.parametrize("tag,lpf", [("nolpf", False), ("lpf50", True)])
def test_eval_windows_match_the_reference_predict_argument(golden_eval, tag, lpf):
    """G5 (the tensor at the reference's model.predict call, Evaluating.py:71-86): f2_gather_windows(centers = NULL,
    normalize = 1) on envelopes that are within 1e-13 of the reference's own (the oracle chain, pinned by G2 / G3), and
    the windows the device pipeline builds from the samples (filterbank kernel + envelope kernel, float64 FFT)."""
    ctx = _lib.default_context()
    wave = golden_eval["g5_wave"]
    coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, 128, 100))
    nb = int(golden_eval[f"g5_{tag}_shape"][0])
    sel = [0, 1, 7000, nb - 1]
    want = golden_eval[f"g5_{tag}_windows"][..., 0]
    sums = golden_eval[f"g5_{tag}_window_sums"]
    env = orc.filter_and_envelope(wave, coefs, lpf, 50)
    out = np.empty((nb, 11, 128), np.float32)
    ctx.gather_windows(env, 128, 16000, None, nb, 5, 160, True, out, _lib.MEM_HOST)
    np.testing.assert_allclose(out[sel], want, rtol=0, atol=1e-7)
    np.testing.assert_allclose(out.reshape(nb, -1).astype(np.float64).sum(axis=1), sums, rtol=1e-6)
    # the same from the device's own envelopes
    env_d = np.empty(128 * 16000)
    ctx.filterbank_envelope_fused(wave, _lib.WAVE_I16, np.array([0, 16000], np.int64), coefs, 1, 128, lpf, 50.0, _lib.FFT_F64,
                                  env_d, None, _lib.MEM_HOST)
    ctx.gather_windows(env_d.reshape(128, 16000), 128, 16000, None, nb, 5, 160, True, out, _lib.MEM_HOST)
    np.testing.assert_allclose(out[sel], want, rtol=0, atol=1e-7)
    np.testing.assert_allclose(out.reshape(nb, -1).astype(np.float64).sum(axis=1), sums, rtol=1e-6)


@pytest.mark.parametrize("rows,channels,n", [(11, 128, 257), (11, 64, 40), (13, 40, 33), (10, 100, 19), (11, 190, 9)])
def test_cnn_forward_vs_oracle(rows, channels, n):
    m = F2CNNModel.glorot(7, rows, channels, zero_bias=False)
    x = np.random.default_rng(3).random((n, rows, channels)).astype(np.float32)
    scores, labels = m.predict_labels(x)
    ref = orc.cnn_forward(x, oracle_weights(m))
    assert scores.shape == (n, 2) and scores.dtype == np.float32
    np.testing.assert_allclose(scores, ref, atol=2e-5)
    ref_labels = orc.labels_from_scores(ref)
    decided = np.abs(ref[:, 1] - ref[:, 0]) > 1e-4
    assert decided.mean() > 0.9
    np.testing.assert_array_equal(labels[decided], ref_labels[decided])
    np.testing.assert_array_equal(labels, (scores[:, 1] > scores[:, 0]).astype(np.uint8))
    np.testing.assert_array_equal(m.predict(x[..., None]), scores)       # (n,R,C,1) accepted like Keras


@pytest.mark.parametrize("C,N", [(128, 16000), (40, 5000), (190, 4000), (3, 2100)])
def test_every_sample_windows_blocked_and_per_window_kernels_agree(C, N):
    """`cnn eval` windows (Evaluating.py:71-80 + Training.py:13-28): the two-pass form (logarithm once per sample, blocks of
    32 consecutive windows; option gather_blocked, default) and the one-workgroup-per-window kernel give the same bits, and
    a non-positive sample raises through either."""
    ctx = _lib.default_context()
    env = np.abs(np.random.default_rng(C).standard_normal((C, N))) + 1e-3
    env[C // 2, N // 2:N // 2 + 400] = 0.25                     # a stretch of equal values
    nb = N - 11 * 160
    outs = {}
    try:
        for opt in (1, 0):
            ctx.set_option("gather_blocked", opt)
            out = np.empty((nb, 11, C), np.float32)
            ctx.gather_windows(env, C, N, None, nb, 5, 160, True, out, _lib.MEM_HOST)
            outs[opt] = out
        np.testing.assert_array_equal(outs[1], outs[0])
        np.testing.assert_allclose(outs[1][::97], orc.eval_input_tensor(env)[::97, ..., 0], atol=2e-7)
        bad = env.copy()
        bad[1, 900] = 0.0
        for opt in (1, 0):
            ctx.set_option("gather_blocked", opt)
            with pytest.raises(_lib.F2Error) as e:
                ctx.gather_windows(bad, C, N, None, nb, 5, 160, True, np.empty((nb, 11, C), np.float32), _lib.MEM_HOST)
            assert e.value.code == _lib.F2_ERR_NONPOSITIVE      # -> ValueError("values must all be positive") in the drivers
    finally:
        ctx.set_option("gather_blocked", 1)


@pytest.mark.parametrize("rows,channels", [(11, 128), (13, 40), (10, 100), (11, 67)])
def test_cnn_f32_and_split_fp16_matrix_paths(rows, channels):
    """conv2-conv4 and dense1 run on the fp16 matrix cores with both operands scaled by per-layer powers of two and split in two
    fp16 pieces (three MFMAs per product, float32 accumulation; option cnn_f16x3, default - f2_cnn_split.h) or on the float32
    matrix cores (0); windows of 10 / 11 rows take the weight-stationary kernels by default (option cnn_ws: conv1 on the
    matrix cores too, f2_cnn_ws.hip). All within the tolerance of the oracle (Training.py:93-114); the split paths at the
    float32 rounding level of the float32 one (5e-7 on softmax scores; the bf16 pieces of rounds 3-4 needed 2e-6), same labels
    wherever the oracle's margin is above that. Windows of 13 rows mix the paths: split conv2 and dense1 around float32 conv3 /
    conv4."""
    ctx = _lib.default_context()
    m = F2CNNModel.glorot(7, rows, channels, zero_bias=False)
    x = np.random.default_rng(8).random((700, rows, channels)).astype(np.float32)
    x[:8] = 0.0
    for i in range(1, 8):
        x[i, (i * 5) % rows, (i * 37) % channels] = 1.0 + 0.4 * i       # impulses: every tap / padding edge (inputs up to 4 are safe)
    ref = orc.cnn_forward(x, oracle_weights(m))
    assert ctx.get_option("cnn_f16x3") == 1 and ctx.get_option("cnn_ws") == 1
    assert ctx.get_option("cnn_bf16x3") == 1                            # (the switch's name in rounds 3-4 still reads it)
    if rows in (10, 11):
        h = m.handle(ctx)
        assert ctx.cnn_info(h, "ws_ok") == 1 and ctx.cnn_info(h, "ws_dense_ok") == 1      # f2_cnn_create's self-check passed
        print(f"self-check of the weight-stationary kernels: convolutions {ctx.cnn_info(h, 'ws_check_diff'):.2e}, "
              f"dense1 {ctx.cnn_info(h, 'ws_dense_check_diff'):.2e}")
        assert ctx.cnn_info(h, "ws_check_diff") <= 1e-6 and ctx.cnn_info(h, "ws_dense_check_diff") <= 1e-6
    got = {}
    try:
        # (ws64: the weight-stationary convolutions with the 64-window dense1 kernel of the split path instead of k_dense1_ws)
        for name, sp, ws, wsd in (("f32", 0, 0, 1), ("f16x3", 1, 0, 1), ("ws", 1, 1, 1), ("ws64", 1, 1, 0)):
            ctx.set_option("cnn_f16x3", sp)
            ctx.set_option("cnn_ws", ws)
            ctx.set_option("cnn_ws_dense", wsd)
            got[name] = m.predict(x, ctx)
            np.testing.assert_allclose(got[name], ref, atol=2e-5, err_msg=name)
    finally:
        ctx.set_option("cnn_f16x3", 1)
        ctx.set_option("cnn_ws", 1)
        ctx.set_option("cnn_ws_dense", 1)
    clear = np.abs(ref[:, 1] - ref[:, 0]) > 1e-5
    for name in ("f16x3", "ws", "ws64"):
        print(f"{rows} x {channels}, {name}: max |score - float32 path| {np.abs(got[name] - got['f32']).max():.2e}")
        assert np.abs(got[name] - got["f32"]).max() <= 5e-7, name
        assert not np.array_equal(got[name], got["f32"]), name        # (the option is not a no-op)
        np.testing.assert_array_equal((got[name][:, 1] > got[name][:, 0])[clear], (got["f32"][:, 1] > got["f32"][:, 0])[clear])
    if rows in (10, 11):
        assert not np.array_equal(got["ws"], got["f16x3"])            # the weight-stationary kernels ran (conv1 differs)


def test_cnn_split_path_scales_follow_the_weights():
    """The split-fp16 path scales every layer's operands by powers of two chosen from the weights (f2_cnn_split.h): networks
    whose weights are 64 x smaller / 8 x larger than Glorot's, or whose biases are large, must come out as accurately as the
    float32 matrix path does - both measured against the oracle's float64-accumulating referee."""
    ctx = _lib.default_context()
    rng = np.random.default_rng(21)
    x = rng.random((300, 11, 128)).astype(np.float32)
    base = F2CNNModel.glorot(5, zero_bias=False)

    def rescaled(f2, f4, bmul):
        # ReLU is positively homogeneous: a layer's kernel and bias times f and the next layer's kernel divided by f leave the
        # network's function alone but move the layer's activations by f
        t = {k: v.astype(np.float64) * (bmul if k.endswith("_b") and k != "dense2_b" else 1.0) for k, v in base.tensors.items()}
        t["conv2_w"] *= f2
        t["conv2_b"] *= f2
        t["conv3_w"] /= f2
        t["conv4_w"] *= f4
        t["conv4_b"] *= f4
        t["dense1_w"] /= f4
        return {k: v.astype(np.float32) for k, v in t.items()}
    for tag, t in (("conv2 outputs 64 x smaller", rescaled(1.0 / 64, 1.0, 1.0)), ("conv4 outputs 32 x larger", rescaled(1.0, 32.0, 1.0)),
                   ("both, biases x 10", rescaled(1.0 / 64, 32.0, 10.0))):
        m = F2CNNModel(t)
        got = m.predict(x, ctx)
        try:
            ctx.set_option("cnn_f16x3", 0)
            ref = m.predict(x, ctx)
        finally:
            ctx.set_option("cnn_f16x3", 1)
        assert np.isfinite(got).all(), tag
        truth = orc.cnn_forward_referee(x, dict(m.tensors))                 # float64 accumulation of the same float32 data
        e_split, e_f32 = np.abs(got - truth).max(), np.abs(ref - truth).max()
        print(f"{tag}: max |score - float64 referee|: split path {e_split:.2e}, float32 path {e_f32:.2e}; "
              f"between the two {np.abs(got - ref).max():.2e}")
        assert e_split <= 2.0 * e_f32 + 2e-7, tag                           # as close to the truth as the float32 kernels are


@pytest.mark.parametrize("rows,channels", [(11, 128), (10, 100)])
def test_cnn_weight_stationary_kernels_at_awkward_window_counts(rows, channels):
    """The persistent kernels walk (window, column tile) tasks with a stride of the grid and dense1 takes 2 x 96 windows per
    workgroup, row blocks dealt out in groups of eight: window counts that leave a workgroup zero, one or two tasks, an empty
    second half, a partial last dense tile, exactly a tile, a group of row blocks with one block more or less - against the per-tile
    kernels (the same split arithmetic in conv2-conv4, float32 conv1: within 5e-7) and the oracle."""
    ctx = _lib.default_context()
    m = F2CNNModel.glorot(3, rows, channels, zero_bias=False)
    rng = np.random.default_rng(12)
    for n in (1, 2, 63, 95, 96, 97, 191, 192, 193, 300, 385, 1025, 1535, 1537):
        x = rng.random((n, rows, channels)).astype(np.float32)
        got = m.predict(x, ctx)
        try:
            ctx.set_option("cnn_ws", 0)
            ref = m.predict(x, ctx)
        finally:
            ctx.set_option("cnn_ws", 1)
        assert np.abs(got - ref).max() <= 5e-7, n
        if n <= 97:
            np.testing.assert_allclose(got, orc.cnn_forward(x, oracle_weights(m)), atol=2e-5)


def test_cnn_structured_inputs():
    # a shifted impulse probes every tap / padding edge of the conv stack; zeros probe the biases
    m = F2CNNModel.glorot(11, zero_bias=False)
    x = np.zeros((24, 11, 128), np.float32)
    for i in range(1, 24):
        x[i, (i * 5) % 11, (i * 37) % 128] = 1.0 + i
    np.testing.assert_allclose(m.predict(x), orc.cnn_forward(x, oracle_weights(m)), atol=2e-5)


def test_cnn_chunking_is_invisible():
    m = F2CNNModel.glorot(5)
    x = np.random.default_rng(4).random((4096 + 70, 11, 128)).astype(np.float32)
    s = m.predict(x)
    np.testing.assert_array_equal(s[:64], m.predict(x[:64]))
    np.testing.assert_array_equal(s[-70:], m.predict(x[-70:]))


@pytest.mark.parametrize("rows,channels,n", [(11, 128, 1024), (13, 40, 300), (10, 100, 300), (11, 64, 300)])
def test_torch_state_dict_conversion(rows, channels, n):
    """Independent evidence for the (unpinned) CNN oracle: the same network evaluated by PyTorch on the CPU
    (Conv2d / max_pool2d / Linear, NCHW) against the oracle restatement AND against K4, >= 1000 windows of the
    reference's 11 x 128 shape plus other row / channel counts."""
    import torch
    torch.manual_seed(0)
    hp2, wp2 = ((rows - 2) // 2 - 2) // 2, ((channels - 2) // 2 - 2) // 2
    net = torch.nn.ModuleDict({
        "conv1": torch.nn.Conv2d(1, 32, 3, padding=1), "conv2": torch.nn.Conv2d(32, 32, 3),
        "conv3": torch.nn.Conv2d(32, 64, 3, padding=1), "conv4": torch.nn.Conv2d(64, 64, 3),
        "dense1": torch.nn.Linear(64 * hp2 * wp2, 516), "dense2": torch.nn.Linear(516, 2)})
    x = torch.rand(n, 1, rows, channels)
    with torch.no_grad():
        h = torch.relu(net["conv2"](torch.relu(net["conv1"](x))))
        h = torch.nn.functional.max_pool2d(h, 2)
        h = torch.relu(net["conv4"](torch.relu(net["conv3"](h))))
        h = torch.nn.functional.max_pool2d(h, 2).flatten(1)
        ref = torch.softmax(net["dense2"](torch.relu(net["dense1"](h))), dim=1).numpy()
    m = F2CNNModel.from_torch_state_dict(net.state_dict(), rows, channels)
    got = m.predict(x[:, 0].numpy())
    np.testing.assert_allclose(got, ref, atol=2e-5)
    o = orc.cnn_forward(x[:, 0].numpy(), oracle_weights(m))
    np.testing.assert_allclose(o, ref, atol=2e-5)
    # labels: torch, oracle and K4 agree wherever the float64 referee's margin is above the float32 rounding level
    r = orc.cnn_forward_referee(x[:, 0].numpy(), oracle_weights(m))
    clear = np.abs(r[:, 1] - r[:, 0]) > 2e-5
    for s in (got, o, ref):
        np.testing.assert_array_equal((s[:, 1] > s[:, 0])[clear], (r[:, 1] > r[:, 0])[clear])


def test_eval_pipeline_cfg4_shape(tmp_path, monkeypatch):
    # BASELINE config 4 in miniature: one 0.25 s utterance end to end, labels vs the oracle chain
    monkeypatch.chdir(tmp_path)
    wave = orc.synth_utterance(2028, 4000)
    m = F2CNNModel.glorot(7)
    scores, labels, env = Evaluating.EvaluateOneWavArray(wave, 16000, model=m, LPF=True, CUTOFF=50, return_envelopes=True)
    nb = 4000 - 1760
    assert scores.shape == (nb, 2) and labels.shape == (nb,)
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100))
    env_ref = orc.filter_and_envelope(wave, coefs, True, 50)
    assert chan_relerr(env, env_ref) <= 1e-5
    ref = orc.cnn_forward(orc.eval_input_tensor(env_ref), oracle_weights(m))
    np.testing.assert_allclose(scores, ref, atol=5e-4)
    # labels: identical, or a tie at the float32-FFT pipeline's rounding level for the float64 referee
    # (same rule as tests/test_gpu_cfg4_labels.py, which runs all 113 920 windows of the full configuration)
    differ = np.flatnonzero(labels != orc.labels_from_scores(ref))
    if len(differ):
        r = orc.cnn_forward_referee(orc.eval_input_tensor(env_ref)[differ], oracle_weights(m))
        assert np.abs(r[:, 1] - r[:, 0]).max() <= 1e-3
    assert len(differ) <= 0.01 * len(labels)
    # too short for a single window: no scores, no error
    s2, l2 = Evaluating.EvaluateOneWavArray(wave[:1700], 16000, model=m)
    assert s2.shape == (0, 2) and l2.shape == (0,)


def test_eval_pipeline_long_utterance(tmp_path, monkeypatch):
    # a 2.5 s utterance goes through the long-row envelope path inside f2_eval_utterance
    monkeypatch.chdir(tmp_path)
    wave = orc.synth_utterance(7, 40000)
    m = F2CNNModel.glorot(3)
    scores, labels, env = Evaluating.EvaluateOneWavArray(wave, 16000, model=m, LPF=False, return_envelopes=True)
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100))
    assert chan_relerr(env, orc.filter_and_envelope(wave, coefs, False)) <= 1e-5
    assert scores.shape == (40000 - 1760, 2)
    np.testing.assert_array_equal(labels, (scores[:, 1] > scores[:, 0]).astype(np.uint8))
    ref = orc.cnn_forward(orc.eval_input_tensor(env)[::97], oracle_weights(m))
    np.testing.assert_allclose(scores[::97], ref, atol=5e-4)
