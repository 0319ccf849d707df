"""The file-level drop-in on the GPU: `prepare filter` -> `prepare envelope` -> `prepare input` -> `cnn eval`
through the CLI, with the reference's directory layout, file names and .npy formats."""
import os

import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import chan_relerr
from f2cnn_amd import cli, config, wavio
from f2cnn_amd.model import F2CNNModel

pytestmark = pytest.mark.gpu


def npy_header(path):
    with open(path, "rb") as f:
        assert f.read(6) == b"\x93NUMPY"
        major, minor = f.read(1)[0], f.read(1)[0]
        hlen = int.from_bytes(f.read(2), "little")
        return (major, minor), f.read(hlen).decode()


def test_prepare_and_eval_cli(tmp_path, monkeypatch, capsys):
    from scipy.io import wavfile
    monkeypatch.chdir(tmp_path)
    config.write_default()
    lens = {"TEST/DR1.FAAA0.SA1": 4000, "TEST/DR1.FBBB0.SX2": 2500, "TRAIN/DR2.MCCC0.SI3": 4097}
    waves = {}
    for i, (key, n) in enumerate(lens.items()):
        os.makedirs(os.path.join("resources", "f2cnn", os.path.dirname(key)), exist_ok=True)
        waves[key] = orc.synth_utterance(100 + i, n)
        path = os.path.join("resources", "f2cnn", key + ".WAV")
        if i == 1:
            wavfile.write(path, 16000, waves[key])        # RIFF
        else:
            wavio.write_sphere(path, 16000, waves[key])   # NIST SPHERE, like TIMIT
    assert cli.main(["prepare", "filter"]) == 0
    assert cli.main(["prepare", "envelope", "--cutoff", "50"]) == 0
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100))
    for key, n in lens.items():
        base = os.path.join("resources", "f2cnn", key)
        ver, hdr = npy_header(base + ".GFB.npy")
        assert ver == (1, 0) and "'<f8'" in hdr and "False" in hdr and f"(128, {n})" in hdr
        gfb, env = np.load(base + ".GFB.npy"), np.load(base + ".ENV1.npy")
        assert env.shape == (128, n) and env.dtype == np.float64
        ref = orc.erb_filterbank(waves[key], coefs)
        assert chan_relerr(gfb, ref) <= 1e-10
        assert chan_relerr(env, orc.extract_envelope_from_matrix(ref, True, 50)) <= 1e-5
    # prepare input
    os.makedirs("trainingData")
    rows = [("TRAIN", "DR2", "MCCC0", "SI3", "aa", 900), ("TEST", "DR1", "FAAA0", "SA1", "iy", 800),
            ("TEST", "DR1", "FBBB0", "SX2", "eh", 1600), ("TEST", "DR1", "FAAA0", "SA1", "ae", 3199)]
    with open("trainingData/label_data.csv", "w") as f:
        for r in rows:
            f.write(",".join(map(str, r)) + ",0.5,0.01,1\n")
    assert cli.main(["prepare", "input", "--cutoff", "50"]) == 0
    x = np.load("trainingData/input_data_LPF50.npy")
    assert x.shape == (4, 11, 128) and x.dtype == np.float32
    np.testing.assert_array_equal(x, np.load("trainingData/last_input_data.npy"))
    order = [("TEST/DR1.FAAA0.SA1", 800), ("TEST/DR1.FAAA0.SA1", 3199), ("TEST/DR1.FBBB0.SX2", 1600),
             ("TRAIN/DR2.MCCC0.SI3", 900)]                      # sorted(file), CSV order inside a file
    for e, (key, tp) in enumerate(order):
        env = np.load(os.path.join("resources", "f2cnn", key + ".ENV1.npy"))
        np.testing.assert_array_equal(x[e], orc.gather_windows(env, [tp])[0].astype(np.float32))
    # the same with two ranks (two processes on this GPU, F2CNN_RANK / F2CNN_WORLD): each gathers its files r::2 into the
    # one pre-sized .npy; byte-identical result, backup copy included, no marker files left behind
    import subprocess
    import sys
    single = open("trainingData/input_data_LPF50.npy", "rb").read()
    os.remove("trainingData/input_data_LPF50.npy")
    os.remove("trainingData/last_input_data.npy")
    procs = [subprocess.Popen([sys.executable, "-m", "f2cnn_amd", "prepare", "input", "--cutoff", "50"],
                              env=dict(os.environ, F2CNN_RANK=str(r), F2CNN_WORLD="2", F2CNN_DEVICE="0",
                                       PYTHONPATH=os.pathsep.join(sys.path)),
                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for r in (1, 0)]
    for pr in procs:
        _, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, err.decode()[-2000:]
    assert open("trainingData/input_data_LPF50.npy", "rb").read() == single
    assert open("trainingData/last_input_data.npy", "rb").read() == single
    assert sorted(os.listdir("trainingData")) == ["input_data_LPF50.npy", "label_data.csv", "last_input_data.npy"]
    # cnn eval
    F2CNNModel.glorot(7).save("last_trained_model.npz")
    wav = os.path.join("resources", "f2cnn", "TEST", "DR1.FAAA0.SA1.WAV")
    assert cli.main(["cnn", "eval", "--file", wav, "--lpf", "50", "--model", "last_trained_model.npz"]) == 0
    res = np.load(os.path.join("resources", "f2cnn", "TEST", "DR1.FAAA0.SA1.F2CNN.npz"))
    assert res["scores"].shape == (4000 - 1760, 2) and res["labels"].shape == (4000 - 1760,)
    out = capsys.readouterr().out
    assert "Total time:" in out and "done !" in out


def test_evalnoise_and_evalrand(tmp_path, monkeypatch):
    from f2cnn_amd.scripts.CNN import Evaluating
    monkeypatch.chdir(tmp_path)
    config.write_default()
    os.makedirs(os.path.join("resources", "f2cnn", "TEST"))
    paths = []
    for i, n in enumerate((3000, 2600)):
        p = os.path.join("resources", "f2cnn", "TEST", f"DR1.S{i}.SA1.WAV")
        wavio.write_sphere(p, 16000, orc.synth_utterance(300 + i, n))
        paths.append(p)
    m = F2CNNModel.glorot(7)
    m.save("last_trained_model.npz")
    # evalnoise: the float64 noisy waveform goes through the float64-input filterbank kernel
    rng = np.random.RandomState(1)
    scores, labels = Evaluating.EvaluateWithNoise(paths[0], model=m, SNRdB=-3, rng=rng)
    noisy_rate, noisy = wavio.read_audio(os.path.join("OutputWavFiles", "addedNoise", "DR1.S0.SA1-3dB.WAV"))
    assert noisy_rate == 16000 and noisy.dtype == np.float64 and noisy.shape == (3000,)
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100))
    env = orc.filter_and_envelope(noisy, coefs, False)
    ref = orc.cnn_forward(orc.eval_input_tensor(env), dict(m.tensors))
    np.testing.assert_allclose(scores, ref, atol=5e-4)
    decided = np.abs(ref[:, 1] - ref[:, 0]) > 2e-3
    np.testing.assert_array_equal(labels[decided], orc.labels_from_scores(ref)[decided])
    # evalrand through the CLI: both files, model loaded once
    assert cli.main(["cnn", "evalrand", "--model", "last_trained_model.npz"]) == 0
    for p in paths:
        assert os.path.exists(os.path.splitext(p)[0] + ".F2CNN.npz")


def test_prepare_features_one_pass_equals_two_commands(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    config.write_default()
    os.makedirs(os.path.join("resources", "f2cnn", "TRAIN"))
    names = []
    for i, n in enumerate((2000, 40000, 1234, 16000, 999)):      # incl. a long row and odd lengths
        p = os.path.join("resources", "f2cnn", "TRAIN", f"DR3.M{i}.SX{i}.WAV")
        wavio.write_sphere(p, 16000, orc.synth_utterance(500 + i, n))
        names.append(os.path.splitext(p)[0])
    assert cli.main(["prepare", "features", "--cutoff", "50"]) == 0
    one = {b: (np.load(b + ".GFB.npy"), np.load(b + ".ENV1.npy")) for b in names}
    for b in names:
        os.remove(b + ".GFB.npy")
        os.remove(b + ".ENV1.npy")
    assert cli.main(["prepare", "filter"]) == 0 and cli.main(["prepare", "envelope", "-c", "50"]) == 0
    for b in names:
        gfb, env = np.load(b + ".GFB.npy"), np.load(b + ".ENV1.npy")
        np.testing.assert_array_equal(gfb, one[b][0])
        assert chan_relerr(one[b][1], env) <= 1e-12        # same float64 hand-off in both routes


def test_batched_eval_matches_one_at_a_time(tmp_path, monkeypatch, capsys):
    """f2_eval_batch (EvaluateWavArrays): ragged batch, incl. an utterance too short to give a window."""
    from f2cnn_amd.scripts.CNN import Evaluating
    monkeypatch.chdir(tmp_path)
    config.write_default()
    m = F2CNNModel.glorot(11)
    waves = [orc.synth_utterance(900 + i, n) for i, n in enumerate((3000, 1700, 2500, 40000, 1761))]
    outs = Evaluating.EvaluateWavArrays(waves, 16000, model=m)
    assert [len(l) for _, l in outs] == [3000 - 1760, 0, 2500 - 1760, 40000 - 1760, 1]
    for w, (scores, labels) in zip(waves, outs):
        s1, l1 = Evaluating.EvaluateOneWavArray(w, 16000, model=m)
        np.testing.assert_allclose(scores, s1, atol=2e-5)
        decided = np.abs(s1[:, 1] - s1[:, 0]) > 1e-4
        np.testing.assert_array_equal(labels[decided], l1[decided])
    # with the low-pass, float64 waves (the evalnoise input type)
    wf = [w.astype(np.float64) * 0.5 for w in waves[:3]]
    outs = Evaluating.EvaluateWavArrays(wf, 16000, model=m, LPF=True, CUTOFF=50)
    for w, (scores, labels) in zip(wf, outs):
        s1, l1 = Evaluating.EvaluateOneWavArray(w, 16000, model=m, LPF=True, CUTOFF=50)
        np.testing.assert_allclose(scores, s1, atol=2e-5)
    with pytest.raises(ValueError, match="sample type"):
        Evaluating.EvaluateWavArrays([waves[0], waves[1].astype(np.float64)], 16000, model=m)
    capsys.readouterr()


def test_resume_metrics_and_unreadable_files(tmp_path, monkeypatch, capsys):
    """`prepare filter / envelope / features` with --skip-existing and --metrics; a file that cannot be read does not
    stop the others and turns the exit status into 2."""
    import json
    monkeypatch.chdir(tmp_path)
    config.write_default()
    os.makedirs(os.path.join("resources", "f2cnn", "TEST"))
    names = []
    for i, n in enumerate((3000, 16000, 40000, 2222, 5000, 1000)):
        p = os.path.join("resources", "f2cnn", "TEST", f"DR2.F{i}.SA{i}.WAV")
        wavio.write_sphere(p, 16000, orc.synth_utterance(700 + i, n))
        names.append(os.path.splitext(p)[0])
    bad = os.path.join("resources", "f2cnn", "TEST", "DR2.BAD.SA9.WAV")
    open(bad, "wb").write(b"NIST_1A\n   1024\nsample_coding -s26 pcm,embedded-shorten-v2.00\nend_head\n".ljust(1024) + b"\0" * 64)
    assert cli.main(["prepare", "filter", "--metrics", "f.json"]) == 2          # six files done, one reported
    m = json.load(open("f.json"))
    assert m["files"] == 6 and m["files_failed"] == 1 and m["failed"][0][0] == bad
    assert abs(m["audio_seconds"] - sum((3000, 16000, 40000, 2222, 5000, 1000)) / 16000) < 1e-3 and m["audio_s_per_s"] > 0
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100))
    for b, seed_n in zip(names, ((700, 3000), (701, 16000), (702, 40000))):
        assert chan_relerr(np.load(b + ".GFB.npy"), orc.erb_filterbank(orc.synth_utterance(*seed_n), coefs)) <= 1e-9
    os.remove(bad)
    # resume: nothing to do, then one output removed -> exactly that file again
    stamp = {b: os.path.getmtime(b + ".GFB.npy") for b in names}
    assert cli.main(["prepare", "filter", "--skip-existing", "--metrics", "f.json"]) == 0
    m = json.load(open("f.json"))
    assert m["files"] == 0 and m["files_skipped"] == 6
    os.remove(names[2] + ".GFB.npy")
    assert cli.main(["prepare", "filter", "--skip-existing", "--metrics", "f.json"]) == 0
    m = json.load(open("f.json"))
    assert m["files"] == 1 and m["files_skipped"] == 5
    assert all(os.path.getmtime(b + ".GFB.npy") == stamp[b] for b in names if b != names[2])
    # envelopes: a truncated .GFB.npy among the inputs
    raw = open(names[4] + ".GFB.npy", "rb").read()
    open(names[4] + ".GFB.npy", "wb").write(raw[:len(raw) // 2])
    assert cli.main(["prepare", "envelope", "-c", "50", "--metrics", "e.json"]) == 2
    m = json.load(open("e.json"))
    assert m["files"] == 5 and m["files_failed"] == 1 and m["failed"][0][0].endswith("F4.SA4.GFB.npy")
    ref = orc.filter_and_envelope(orc.synth_utterance(702, 40000), coefs, True, 50)
    assert chan_relerr(np.load(names[2] + ".ENV1.npy"), ref) <= 1e-5
    assert not os.path.exists(names[4] + ".ENV1.npy")
    # resume after a killed run: a half-written output does not count as done (the resume checks the header against the
    # file size)
    raw_env = open(names[1] + ".ENV1.npy", "rb").read()
    open(names[1] + ".ENV1.npy", "wb").write(raw_env[:len(raw_env) // 3])
    assert cli.main(["prepare", "envelope", "-c", "50", "--skip-existing", "--metrics", "e.json"]) == 2
    m = json.load(open("e.json"))
    assert m["files"] == 1 and m["files_skipped"] == 4 and m["files_failed"] == 1
    assert open(names[1] + ".ENV1.npy", "rb").read() == raw_env
    # resume with another cutoff: envelopes made with the old one are not kept
    assert cli.main(["prepare", "envelope", "-c", "100", "--skip-existing", "--metrics", "e.json"]) == 2
    m = json.load(open("e.json"))
    assert m["files"] == 5 and m["files_skipped"] == 0
    assert chan_relerr(np.load(names[2] + ".ENV1.npy"), orc.filter_and_envelope(orc.synth_utterance(702, 40000), coefs, True, 100)) <= 1e-5
    assert cli.main(["prepare", "envelope", "-c", "50", "--skip-existing"]) == 2     # ... and back, for the lines below
    # one pass with resume: only the file whose envelope is missing (and its .GFB.npy rewritten)
    assert cli.main(["prepare", "features", "-c", "50", "--skip-existing", "--metrics", "p.json"]) == 0
    m = json.load(open("p.json"))
    assert m["files"] == 1 and m["files_skipped"] == 5 and os.path.exists(names[4] + ".ENV1.npy")
    capsys.readouterr()
