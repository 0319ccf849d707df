"""CNN training (reference scripts/CNN/Training.py:31-156). Host logic on CPU (explicit
device="cpu"); the GPU tests train on the card and check the trained weights through HIP kernel K4."""
import os

import numpy as np
import pytest

from f2cnn_amd import cli, config
from f2cnn_amd.model import F2CNNModel, load_model
from f2cnn_amd.scripts.CNN import Training
from oracle import f2cnn_oracle as orc


def synthetic_windows(n, seed, positive=True):
    """Windows whose energy ramps up (label 1, 'rising') or down (label 0) over the 11 time steps."""
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 2, n)
    x = rng.uniform(0.2, 0.4, (n, 11, 128))
    ramp = np.linspace(0.0, 0.5, 11)[:, None]
    x[y == 1] += ramp
    x[y == 0] += ramp[::-1]
    return (x + 1.0 if positive else x), y


def write_training_set(tmp, n_train=64, n_test=32):
    x, y = synthetic_windows(n_train + n_test, 5)
    split = np.array(["TRAIN"] * n_train + ["TEST"] * n_test)
    order = np.random.default_rng(2).permutation(len(y))          # TEST / TRAIN rows interleaved
    x, y, split = x[order], y[order], split[order]
    os.makedirs(os.path.join(tmp, "trainingData"), exist_ok=True)
    np.save(os.path.join(tmp, "trainingData", "last_input_data.npy"), x)
    with open(os.path.join(tmp, "trainingData", "label_data.csv"), "w") as f:
        for i in range(len(y)):
            f.write(f"{split[i]},DR1,FAAA0,SA1,aa,{1000 + i},0.5,0.01,{y[i]}\n")
    return x, y, split


def test_separate_test_train_pairs_rows_by_index(tmp_path):
    x, y, split = write_training_set(str(tmp_path))
    xt, yt, xr, yr = Training.SeparateTestTrain(str(tmp_path / "trainingData" / "last_input_data.npy"),
                                                str(tmp_path / "trainingData" / "label_data.csv"))
    np.testing.assert_array_equal(xt, x[split == "TEST"])
    np.testing.assert_array_equal(yt, y[split == "TEST"])
    np.testing.assert_array_equal(xr, x[split == "TRAIN"])
    np.testing.assert_array_equal(yr, y[split == "TRAIN"])


def test_train_network_history_and_export():
    x, y = synthetic_windows(48, 1, positive=False)
    model, hist = Training.train_network(x[:32], y[:32], x[32:], y[32:], batch_size=16, epochs=2, device="cpu",
                                         seed=3, verbose=0)
    assert set(hist) == {"loss", "acc", "val_loss", "val_acc"} and all(len(v) == 2 for v in hist.values())
    assert isinstance(model, F2CNNModel) and model.tensors["dense1_w"].shape == (1 * 30 * 64, 516)
    assert hist["loss"][1] < hist["loss"][0]                       # RMSprop moves downhill on this toy problem


def test_early_stopping_patience():
    # lr = 0: val_acc never improves by min_delta after the first epoch -> stop after 1 + patience epochs
    x, y = synthetic_windows(24, 2, positive=False)
    _, hist = Training.train_network(x[:16], y[:16], x[16:], y[16:], batch_size=8, epochs=20, device="cpu", seed=0,
                                     verbose=0, lr=0.0)
    assert len(hist["val_acc"]) == 6
    assert len(set(hist["val_acc"])) == 1


def test_rmsprop_matches_keras_update_rule():
    """One step of the optimiser against the Keras 2.2 formula: a = 0.9 a + 0.1 g^2; p -= lr_t g / (sqrt(a) + 1e-7),
    lr_t = lr / (1 + decay * iterations)."""
    import torch
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64))
    opt = torch.optim.RMSprop([p], lr=1e-4, alpha=0.9, eps=1e-7)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 1.0 / (1.0 + 1e-6 * it))
    a, ref = np.zeros(3), np.array([1.0, -2.0, 0.5])
    for it in range(3):
        g = np.array([0.3, -0.1, 2.0]) * (it + 1)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        sched.step()
        a = 0.9 * a + 0.1 * g * g
        ref = ref - (1e-4 / (1.0 + 1e-6 * it)) * g / (np.sqrt(a) + 1e-7)
        np.testing.assert_allclose(p.detach().numpy(), ref, rtol=1e-12)


def test_training_needs_a_gpu_unless_overridden():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x, y = synthetic_windows(8, 0, positive=False)
    with pytest.raises(RuntimeError, match="needs a GPU"):
        Training.train_network(x, y, x, y, epochs=1)


def test_model_keeps_reference_file_name(tmp_path):
    m = F2CNNModel.glorot(3)
    m.save(tmp_path / "last_trained_model")
    assert (tmp_path / "last_trained_model").is_file() and not (tmp_path / "last_trained_model.npz").exists()
    back = load_model(tmp_path / "last_trained_model")
    np.testing.assert_array_equal(back.tensors["conv3_w"], m.tensors["conv3_w"])
    m.save(tmp_path / "other.npz")
    assert load_model(str(tmp_path / "other")).rows == 11          # '.npz' found when the bare name is given


@pytest.mark.gpu
def test_normalize_batch_matches_oracle():
    x, _ = synthetic_windows(40, 9)
    got = Training.normalizeInputBatch(x)
    ref = np.stack([orc.normalize_input(w) for w in x]).astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    bad = x.copy()
    bad[7, 3, 5] = 0.0
    with pytest.raises(ValueError, match="positive"):
        Training.normalizeInputBatch(bad)


@pytest.mark.gpu
def test_cnn_train_cli_and_trained_weights_through_k4(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    config.write_default(batch_size=16, epochs=12)
    x, y, split = write_training_set(str(tmp_path), n_train=192, n_test=64)
    assert cli.main(["cnn", "train"]) == 0
    out = capsys.readouterr().out
    assert "Test accuracy:" in out and os.path.isfile("last_trained_model") and os.path.isfile("last_trained_model_results.json")
    # a trained network has real decision boundaries: K4 must reproduce the float32 oracle's labels on them
    model = load_model("last_trained_model")
    xt = Training.normalizeInputBatch(x[split == "TEST"])
    scores, labels = model.predict_labels(xt)
    ref = orc.cnn_forward(xt, dict(model.tensors))
    np.testing.assert_allclose(scores, ref, atol=2e-5)
    decided = np.abs(ref[:, 1] - ref[:, 0]) > 1e-4
    np.testing.assert_array_equal(labels[decided], orc.labels_from_scores(ref)[decided])
    assert decided.mean() > 0.9
    assert (labels == y[split == "TEST"]).mean() >= 0.9           # the toy task is separable
