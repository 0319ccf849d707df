#!/usr/bin/env python3
"""Golden vectors for the label-data readers: synthetic .FB / .PHN files parsed by the REFERENCE's own
FBFileReader / PHNFileReader (imported read-only from /root/reference). LabelDataGenerator itself cannot be
imported (needs `sphfile`), so ExtractLabel is pinned through these readers plus the oracle restatement.

    python tests/golden/make_golden_labels.py
"""
import os
import struct
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.environ.get("F2CNN_REFERENCE", "/root/reference"))
HERE = os.path.dirname(os.path.abspath(__file__))
from scripts.processing import FBFileReader as ref_fb      # noqa: E402
from scripts.processing import PHNFileReader as ref_phn    # noqa: E402


def synth_fb(n_frames, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n_frames)
    f = np.stack([0.5 + 0.1 * np.sin(t / 7.0), 1.5 + 0.4 * np.sin(t / 9.0 + 1), 2.5 + 0.2 * np.cos(t / 5.0),
                  3.5 + 0.1 * np.sin(t / 3.0)], axis=1) + rng.normal(scale=0.01, size=(n_frames, 4))
    b = 0.05 + 0.02 * rng.random((n_frames, 4))
    return np.hstack([f, b]).astype(np.float32)


def fb_bytes(frames):
    return struct.pack('>iihh', frames.shape[0], 100000, 32, 9) + frames.astype('>f4').tobytes()


PHN_TEXT = "0 2000 h#\n2000 4100 sh\n4100 6000 iy\n6000 7400 pau\n7400 9000 hv\n9000 12000 ae\n12000 16000 h#\n"


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, n, seed in (("a", 100, 1), ("b", 263, 2)):
            frames = synth_fb(n, seed)
            path = os.path.join(tmp, tag + ".FB")
            with open(path, "wb") as f:
                f.write(fb_bytes(frames))
            m, period = ref_fb.ExtractFBFile(path)
            out[f"fb_{tag}_frames_f32"] = frames
            out[f"fb_{tag}_matrix"] = m
            out[f"fb_{tag}_period"] = np.array(period)
            f2, _ = ref_fb.GetFormantFrequencies(path, 2)
            out[f"fb_{tag}_f2"] = f2
            out[f"fb_{tag}_around"] = np.array(ref_fb.GetFromantFrequenciesAround(f2, 4800, 5, 160.0))
        # rounding rule of the reader (built-in round(d * 1000, 2) per value): 500 frames of random values plus the
        # float32 numbers closest to two-decimal ties x.xx5 Hz, from both sides
        rng = np.random.default_rng(3)
        ties = (rng.integers(20000, 400000, size=2000) + 0.5) / 100.0 / 1000.0          # kHz
        t32 = ties.astype(np.float32)
        near = np.concatenate([np.nextafter(t32[:667], np.float32(0)), t32[667:1334], np.nextafter(t32[1334:], np.float32(10))])
        frames = np.concatenate([near, rng.uniform(0.05, 4.0, size=2000).astype(np.float32)]).astype(np.float32).reshape(-1, 8)
        path = os.path.join(tmp, "t.FB")
        with open(path, "wb") as f:
            f.write(fb_bytes(frames))
        out["fb_t_frames_f32"] = frames
        out["fb_t_matrix"] = ref_fb.ExtractFBFile(path)[0]
        phn = os.path.join(tmp, "x.PHN")
        with open(phn, "w") as f:
            f.write(PHN_TEXT)
        ph = ref_phn.ExtractPhonemes(phn)
        out["phn_text"] = np.array(PHN_TEXT)
        out["phn_names"] = np.array([p[0] for p in ph])
        out["phn_bounds"] = np.array([[p[1], p[2]] for p in ph])
        pts = [0, 1999, 2000, 2001, 4100, 7000, 9000, 15999, 16001, 50000]
        out["phn_query_points"] = np.array(pts)
        out["phn_query_answers"] = np.array([ref_phn.GetPhonemeFromArrayAt(ph, t) for t in pts])
        out["phn_silents"] = np.array(ref_phn.SILENTS)
        out["missing_fb"] = np.array(int(ref_fb.ExtractFBFile(os.path.join(tmp, "none.FB"))[0] is None))
        out["missing_phn"] = np.array(int(ref_phn.ExtractPhonemes(os.path.join(tmp, "none.PHN")) is None))
    path = os.path.join(HERE, "f2cnn_golden_labels.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
