#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only, from
/root/reference) in this container. Only data (inputs + the reference's outputs) is written; no
reference source is copied. The reference does not exist on the GPU box, so the .npz files are
committed and this script documents how they were made.

    python tests/golden/make_golden.py          # needs /root/reference; writes f2cnn_golden.npz (G1-G4, G6)
    python tests/golden/make_golden.py eval     # writes f2cnn_golden_eval.npz (G5) only

Imports used (all import cleanly here): gammatone.filters,
scripts.processing.EnvelopeExtraction, scripts.processing.InputGenerator, scripts.CNN.Training
(normalizeInput only).

G5: scripts.CNN.Evaluating imports `sphfile` (through GammatoneFiltering.py:21) and `keras`
(Evaluating.py:84), neither of which exists in this image. Two stand-in MODULES are written into a scratch directory
(never into the repo): `sphfile.py` whose SPHFile refuses to be used (the RIFF branch of GetArrayFromWAV never reaches it)
and a `keras` package whose `models.load_model(...).predict(x)` stores x and stops the run. Nothing of the reference
executes differently before that point: its unmodified EvaluateOneWavFile reads the RIFF file, filters, extracts the
envelopes, gathers and normalises every window and calls predict (Evaluating.py:42-86) - the tensor it hands over is
the fixture. (The stand-ins replace the absent third-party packages, not reference code; the CNN arithmetic itself stays
unpinned - there is no Keras to run.)
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("F2CNN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from gammatone import filters as ref_filters                      # noqa: E402
from scripts.processing import EnvelopeExtraction as ref_env      # noqa: E402
from scripts.processing import InputGenerator as ref_input        # noqa: E402
from scripts.CNN import Training as ref_training                  # noqa: E402


def noise(seed, n, sigma=3000.0):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.standard_normal(n) * sigma), -32768, 32767).astype(np.int16)


def main():
    out = {}
    # G1: filter design
    for C in (8, 64, 128):
        cf = ref_filters.centre_freqs(16000, C, 100)
        out[f"g1_cf_{C}"] = cf
        out[f"g1_coefs_{C}"] = ref_filters.make_erb_filters(16000, cf)
    co8, co128 = out["g1_coefs_8"], out["g1_coefs_128"]

    # G2/G3: filterbank + envelopes (LPF off / 50 / 100)
    def case(tag, wave, coefs, rows, lpfs):
        gfb = ref_filters.erb_filterbank(wave, coefs)
        out[f"g2_{tag}_wave"] = wave
        out[f"g2_{tag}_rows"] = np.asarray(rows)
        out[f"g2_{tag}_gfb"] = gfb[rows]
        for lpf in lpfs:
            env = ref_env.ExtractEnvelopeFromMatrix(gfb[rows], bool(lpf), lpf or 100)
            out[f"g3_{tag}_env_lpf{lpf}"] = env

    case("n1000_c8", noise(1234, 1000), co8, list(range(8)), (0, 50, 100))
    case("n4096_c8", noise(1235, 4096), co8, [0, 7], (0, 50))
    case("n4097_c8", noise(1236, 4097), co8, [0, 7], (0, 50))
    case("n16000_c128", noise(1234, 16000), co128, [0, 127], (0, 50))
    case("n2500_c128", noise(1237, 2500), co128, [1, 64, 126], (0, 50))
    imp = np.zeros(512, dtype=np.int16)
    imp[0] = 1
    case("impulse_c8", imp, co8, list(range(8)), (0,))
    sine = np.round(10000 * np.sin(2 * np.pi * 1000 * np.arange(2000) / 16000)).astype(np.int16)
    case("sine1k_c8", sine, co8, [0, 4, 7], (0, 50))
    # float64 wave (the `evalnoise` path feeds a float64 array, Evaluating.py:199-200)
    fw = noise(1238, 1500).astype(np.float64) + np.random.default_rng(5).normal(scale=500.0, size=1500)
    case("n1500_f64_c8", fw, co8, [0, 3, 7], (0,))

    # G6: normalizeInput
    rng = np.random.default_rng(99)
    w64 = rng.random((11, 128)) * 50 + 1e-3
    out["g6_in_f64"] = w64
    out["g6_out_f64"] = ref_training.normalizeInput(w64.copy())
    w32 = w64.astype(np.float32).reshape(11, 128, 1)
    out["g6_in_f32"] = w32
    out["g6_out_f32"] = ref_training.normalizeInput(w32.copy())
    out["g6_const_out"] = ref_training.normalizeInput(np.full((11, 128), 3.25))
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            ref_training.normalizeInput(np.zeros((11, 128)))
        out["g6_nonpositive_raises"] = np.array(0)
    except ValueError:
        out["g6_nonpositive_raises"] = np.array(1)

    # G4: GenerateInputData on a crafted 3-file / 12-row CSV. ENV1 inputs are reproducible from seeds
    # (tests rebuild them), only the reference's output tensor is stored.
    rows = [("TRAIN", "DR2", "MABC0", "SX10", "aa", 8000), ("TEST", "DR1", "FXYZ0", "SA1", "iy", 800),
            ("TRAIN", "DR2", "MABC0", "SX10", "eh", 960), ("TEST", "DR1", "FXYZ0", "SA1", "ae", 15199),
            ("TEST", "DR1", "FAAA0", "SI5", "uw", 5000), ("TRAIN", "DR2", "MABC0", "SX10", "ih", 15199),
            ("TEST", "DR1", "FXYZ0", "SA1", "ao", 960), ("TEST", "DR1", "FAAA0", "SI5", "er", 801),
            ("TEST", "DR1", "FAAA0", "SI5", "ey", 12345), ("TRAIN", "DR2", "MABC0", "SX10", "ay", 800),
            ("TEST", "DR1", "FXYZ0", "SA1", "ow", 8000), ("TEST", "DR1", "FAAA0", "SI5", "oy", 7777)]
    envseeds = {"TEST/DR1.FXYZ0.SA1": 11, "TEST/DR1.FAAA0.SI5": 12, "TRAIN/DR2.MABC0.SX10": 13}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            os.makedirs("trainingData")
            with open("configF2CNN.conf", "w") as f:
                f.write("[FILTERBANK]\nframerate = 16000\nnchannels = 128\nlow_freq = 100\n\n"
                        "[CNN]\nformant = 2\ncentered = True\nradius = 5\nbatch_size = 32\nepochs = 20\n"
                        "risk = 0.05\nsampling_period = 10000\n")
            for key, seed in envseeds.items():
                os.makedirs(os.path.join("resources", "f2cnn", os.path.dirname(key)), exist_ok=True)
                env = np.random.default_rng(seed).random((128, 16000)) + 0.01
                np.save(os.path.join("resources", "f2cnn", key + ".ENV1.npy"), env)
            with open(os.path.join("trainingData", "label_data.csv"), "w") as f:
                for (s, r, sp, se, ph, tp) in rows:
                    f.write(f"{s},{r},{sp},{se},{ph},{tp},0.5,0.01,1\n")
            with contextlib.redirect_stdout(io.StringIO()):
                ref_input.GenerateInputData(LPF=False, CUTOFF=None)
            out["g4_input_data"] = np.load(os.path.join("trainingData", "input_data_NOLPF.npy"))
            out["g4_last_input_equal"] = np.array(int(np.array_equal(
                out["g4_input_data"], np.load(os.path.join("trainingData", "last_input_data.npy")))))
        finally:
            os.chdir(cwd)
    out["g4_csv_rows"] = np.array([",".join(map(str, r)) for r in rows])
    out["g4_env_keys"] = np.array(list(envseeds.keys()))
    out["g4_env_seeds"] = np.array(list(envseeds.values()))

    path = os.path.join(HERE, "f2cnn_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


def g5_eval_tensor():
    """The (nb, 11, 128, 1) float64 tensor at the reference's `predict` call for one 1 s RIFF file, LPF off and 50 Hz."""
    import hashlib
    import importlib
    from scipy.io import wavfile
    out = {}
    wave = noise(2028, 16000)          # utterance 0 of BASELINE cfg4 (seed 2028)
    out["g5_wave"] = wave
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        stubs = os.path.join(tmp, "absent_packages")
        os.makedirs(os.path.join(stubs, "keras"))
        with open(os.path.join(stubs, "sphfile.py"), "w") as f:
            f.write("class SPHFile:\n    def __init__(self, *a, **k):\n        raise RuntimeError('sphfile is not installed')\n")
        with open(os.path.join(stubs, "keras", "__init__.py"), "w") as f:
            f.write("from . import models, backend\n")
        with open(os.path.join(stubs, "keras", "backend.py"), "w") as f:
            f.write("def clear_session():\n    pass\n")
        with open(os.path.join(stubs, "keras", "models.py"), "w") as f:
            f.write("import numpy\n\n\nclass Reached(Exception):\n    pass\n\n\nclass _Recorder:\n"
                    "    def __init__(self, path):\n        self.path = path\n\n"
                    "    def predict(self, x, verbose=0):\n        numpy.save('predict_argument.npy', x)\n"
                    "        raise Reached(self.path)\n\n\ndef load_model(path):\n    return _Recorder(path)\n")
        sys.path.insert(0, stubs)
        os.environ["MPLBACKEND"] = "Agg"
        os.chdir(tmp)
        try:
            with open("configF2CNN.conf", "w") as f:
                f.write("[FILTERBANK]\nframerate = 16000\nnchannels = 128\nlow_freq = 100\n\n"
                        "[CNN]\nformant = 2\ncentered = True\nradius = 5\nbatch_size = 32\nepochs = 20\n"
                        "risk = 0.05\nsampling_period = 10000\n")
            os.makedirs("resources/f2cnn/TEST")
            wav = "resources/f2cnn/TEST/DR1.FXYZ0.SA1.WAV"
            wavfile.write(wav, 16000, wave)
            ref_eval = importlib.import_module("scripts.CNN.Evaluating")
            reached = importlib.import_module("keras.models").Reached
            for tag, lpf in (("nolpf", False), ("lpf50", True)):
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        ref_eval.EvaluateOneWavFile(wav, LPF=lpf, CUTOFF=50, model="last_trained_model")
                    raise AssertionError("predict was not reached")
                except reached:
                    pass
                x = np.load("predict_argument.npy")
                os.remove("predict_argument.npy")
                assert x.shape == (16000 - 11 * 160, 11, 128, 1) and x.dtype == np.float64, (x.shape, x.dtype)
                nb = x.shape[0]
                out[f"g5_{tag}_windows"] = x[[0, 1, 7000, nb - 1]]
                out[f"g5_{tag}_window_sums"] = x.reshape(nb, -1).sum(axis=1)
                out[f"g5_{tag}_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest())
                out[f"g5_{tag}_shape"] = np.array(x.shape)
        finally:
            os.chdir(cwd)
            sys.path.remove(stubs)
    path = os.path.join(HERE, "f2cnn_golden_eval.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    if sys.argv[1:] == ["eval"]:
        g5_eval_tensor()
    else:
        main()
