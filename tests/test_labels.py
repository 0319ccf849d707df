"""Label-data producer, CPU only: readers against golden vectors produced by the reference's
own FBFileReader / PHNFileReader, ExtractLabel against the oracle restatement, CSV consumed by prepare input."""
import os
import struct
from configparser import ConfigParser

import numpy as np
import pytest

import f2cnn_oracle as orc
from conftest import ROOT
from f2cnn_amd import config, wavio
from f2cnn_amd.scripts.processing import FBFileReader, InputGenerator, LabelDataGenerator, PHNFileReader


@pytest.fixture(scope="module")
def gl():
    return np.load(os.path.join(ROOT, "tests", "golden", "f2cnn_golden_labels.npz"))


def write_fb(path, frames):
    with open(path, "wb") as f:
        f.write(struct.pack('>iihh', frames.shape[0], 100000, 32, 9) + frames.astype('>f4').tobytes())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fb_reader_matches_reference(gl, tag, tmp_path):
    p = str(tmp_path / "x.FB")
    write_fb(p, gl[f"fb_{tag}_frames_f32"])
    for reader in (FBFileReader.ExtractFBFile, orc.extract_fb_file):
        m, period = reader(p)
        assert period == int(gl[f"fb_{tag}_period"]) == 10000
        np.testing.assert_array_equal(m, gl[f"fb_{tag}_matrix"])
    f2, _ = FBFileReader.GetFormantFrequencies(p, 2)
    np.testing.assert_array_equal(f2, gl[f"fb_{tag}_f2"])
    np.testing.assert_array_equal(FBFileReader.GetFromantFrequenciesAround(f2, 4800, 5, 160.0), gl[f"fb_{tag}_around"])
    assert FBFileReader.ExtractFBFile(str(tmp_path / "none.FB")) == (None, 0) and int(gl["missing_fb"]) == 1
    with pytest.raises(SystemExit):
        FBFileReader.GetFromantFrequenciesAround(f2, 100, 5, 160.0)


def test_fb_rounding_is_the_builtin_round(gl, tmp_path):
    """4000 values (float32 neighbours of two-decimal ties + random) against the reference reader's output, and the
    rounding helper against round() on doubles where numpy.round lands on the other side of the tie."""
    p = str(tmp_path / "t.FB")
    write_fb(p, gl["fb_t_frames_f32"])
    assert gl["fb_t_frames_f32"].size == 4000
    for reader in (FBFileReader.ExtractFBFile, orc.extract_fb_file):
        np.testing.assert_array_equal(reader(p)[0], gl["fb_t_matrix"])
    hard = np.array([2.675, 1.005, 0.125, 0.375, 1234.565, 1000.5 / 100, 8.345, -2.675, 2.665, 1e-9, 0.0])
    assert any(np.round(v, 2) != round(float(v), 2) for v in hard)            # the two rules do differ here
    np.testing.assert_array_equal(FBFileReader.round_half_even_decimal(hard, 2), [round(float(v), 2) for v in hard])
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.uniform(0, 4000, 20000), (rng.integers(0, 400000, 20000) + 0.5) / 100])
    np.testing.assert_array_equal(FBFileReader.round_half_even_decimal(v, 2), [round(float(x), 2) for x in v])


def test_label_rows_are_vectorised_and_tie_safe():
    """slopes_and_pvalues against per-row lstsq + pearsonr (the reference's formulation), rounded as the CSV is,
    on windows of two-decimal values where exact rounding ties are frequent."""
    from scipy.stats import pearsonr
    rng = np.random.default_rng(5)
    windows = np.round(1500 + 400 * rng.standard_normal((3000, 1)) + 40 * rng.standard_normal((3000, 11)), 2)
    steps = (800 + 160 * np.arange(3000)).astype(np.float64)
    a, p = LabelDataGenerator.slopes_and_pvalues(windows, steps, 160)
    ties = 0
    for i in range(3000):
        x = steps[i] + 160.0 * (np.arange(11) - 5)
        (ar, br), _, _, _ = np.linalg.lstsq(np.vstack([x, np.ones(11)]).T, windows[i], rcond=None)
        pr = pearsonr(windows[i], ar * x + br)[1]
        assert round(ar, 5) == np.round(a[i], 5) and round(pr, 5) == np.round(p[i], 5), i
        s5 = abs(ar) * 1e5
        ties += abs(s5 - np.floor(s5) - 0.5) < 1e-6
    assert ties > 20        # the fixture does exercise ties
    assert LabelDataGenerator.phonemes_at([("aa", 0, 10), ("bb", 10, 20)], [0, 10, 11, 21]).tolist() == ["aa", "aa", "bb", "h#"]


def test_phn_reader_matches_reference(gl, tmp_path):
    p = str(tmp_path / "x.PHN")
    open(p, "w").write(str(gl["phn_text"]))
    for reader in (PHNFileReader.ExtractPhonemes, orc.extract_phonemes):
        ph = reader(p)
        assert [x[0] for x in ph] == list(gl["phn_names"])
        np.testing.assert_array_equal(np.array([[x[1], x[2]] for x in ph]), gl["phn_bounds"])
    ph = PHNFileReader.ExtractPhonemes(p)
    got = [PHNFileReader.GetPhonemeFromArrayAt(ph, int(t)) for t in gl["phn_query_points"]]
    assert got == list(gl["phn_query_answers"])
    assert PHNFileReader.SILENTS == list(gl["phn_silents"])
    assert PHNFileReader.ExtractPhonemes(str(tmp_path / "none.PHN")) is None and int(gl["missing_phn"]) == 1


def make_corpus(gl, tmp_path):
    os.chdir(tmp_path)
    config.write_default()
    os.makedirs("resources/f2cnn/TEST")
    os.makedirs("resources/f2cnn/TRAIN")
    files = {"resources/f2cnn/TEST/DR1.FAKE0.SA1": ("a", 16000), "resources/f2cnn/TRAIN/DR4.MOCK0.SX9": ("b", 41000)}
    for base, (tag, n) in files.items():
        wavio.write_sphere(base + ".WAV", 16000, orc.synth_utterance(hash(tag) % 1000, n))
        write_fb(base + ".FB", gl[f"fb_{tag}_frames_f32"])
        open(base + ".PHN", "w").write(str(gl["phn_text"]).replace("12000 16000 h#", "12000 30000 eh\n30000 41000 h#") if tag == "b"
                                       else str(gl["phn_text"]))
    return files


def test_extract_label_matches_oracle(gl, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    files = make_corpus(gl, tmp_path)
    cp = ConfigParser()
    cp.read("configF2CNN.conf")
    total = 0
    for base, (tag, n) in files.items():
        got = LabelDataGenerator.ExtractLabel(base + ".WAV", cp)
        f2 = gl[f"fb_{tag}_f2"]
        ids = (os.path.basename(os.path.dirname(base)), *os.path.basename(base).split("."))
        ref = orc.extract_label(n, 16000, f2, orc.extract_phonemes(base + ".PHN"), ids)
        assert got is not None and len(got) == len(ref) > 5
        for g, r in zip(got, ref):
            assert g[:6] == r[:6] and g[8] == r[8]
            assert g[6] == r[6] and g[7] == r[7]
        assert all(row[4] not in PHNFileReader.SILENTS for row in got)
        assert all(row[7] < 0.05 for row in got)
        total += len(got)
    # a file without .FB yields nothing
    wavio.write_sphere("resources/f2cnn/TEST/DR1.NOFB0.SA2.WAV", 16000, orc.synth_utterance(3, 8000))
    assert LabelDataGenerator.ExtractLabel("resources/f2cnn/TEST/DR1.NOFB0.SA2.WAV", cp) is None
    # the CSV written by `prepare label` is what `prepare input` parses
    LabelDataGenerator.GenerateLabelData()
    d = InputGenerator.GetListOfEnvelopeFilesAndTimepoints("trainingData/label_data.csv")
    assert sum(len(v) for v in d.values()) == total
    assert set(d) == {os.path.join("TEST", "DR1.FAKE0.SA1.ENV1.npy"), os.path.join("TRAIN", "DR4.MOCK0.SX9.ENV1.npy")}
    first = open("trainingData/label_data.csv").readline().strip().split(",")
    assert len(first) == 9 and first[0] in ("TEST", "TRAIN") and first[8] in ("0", "1")
