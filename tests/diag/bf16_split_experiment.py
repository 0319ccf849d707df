"""Would a split-bf16 CNN (v_mfma_f32_32x32x16_bf16 on operands split into bf16 pieces, f32 accumulation) keep the
labels of cfg4?  Numerical experiment on the CPU: conv2, conv3, conv4 and dense1 of the
oracle CNN (oracle/f2cnn_oracle.py cnn_forward, after /root/reference/scripts/CNN/Training.py:93-114) are evaluated with
both GEMM operands replaced by sums of bf16 pieces,

    a = a1 + a2 (+ a3),  a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)     (round to nearest even)

and the partial products a MFMA kernel would issue (each exact in float32, summed in float32):

    "3x":  a1 b1 + a1 b2 + a2 b1                         (3 bf16 MFMAs per f32 MFMA's worth of K: 16/3 = 5.3 x its rate)
    "6x":  3x + a2 b2 + a1 b3 + a3 b1                    (6 MFMAs: 2.7 x)

conv1 (VALU, inside conv2's staging) and dense2 + softmax stay float32.  Windows: the cfg4 tensors of the oracle chain
(8 synthetic utterances of 1 s, 14 240 windows each; --utterances picks fewer).  Referee rule of
tests/test_gpu_cfg4_labels.py: labels equal to the float32 oracle's, or the float64 referee's margin of a differing
window is a tie (<= 2e-6); scores within 2e-5.

Run:  python tests/diag/bf16_split_experiment.py [--utterances 8]     (CPU only, a few minutes per utterance set)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import f2cnn_oracle as orc   # noqa: E402


def bf16(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def pieces(x, n):
    out, rest = [], np.asarray(x, dtype=np.float32)
    for _ in range(n):
        p = bf16(rest)
        out.append(p)
        rest = rest - p
    return out


TERMS = {"3x": [(0, 0), (0, 1), (1, 0)], "6x": [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]}


def split_dot(a, b, mode):
    """a (m,k) . b (k,n) with both operands split; float32 accumulation of the partial GEMMs, smallest terms first"""
    if mode == "f32":
        return a.dot(b)
    ap, bp = pieces(a, 3), pieces(b, 3)
    acc = None
    for i, j in reversed(TERMS[mode]):
        t = ap[i].dot(bp[j])
        acc = t if acc is None else acc + t
    return acc


def conv3x3(x, w, b, same, mode):
    if same:
        x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    n, H, W, ci = x.shape
    Ho, Wo = H - 2, W - 2
    # one GEMM over K = 9 ci (the implicit-GEMM form the MFMA kernels use)
    cols = np.concatenate([x[:, dy:dy + Ho, dx:dx + Wo, :] for dy in range(3) for dx in range(3)], axis=-1)
    out = split_dot(cols.reshape(-1, 9 * ci), w.reshape(9 * ci, -1), mode)
    return out.reshape(n, Ho, Wo, -1) + b


def logits(x, w, mode, chunk=512):
    """dense2 outputs before the softmax (float32)"""
    relu = lambda v: np.maximum(v, 0)
    outs = []
    for s in range(0, x.shape[0], chunk):
        h = x[s:s + chunk, :, :, None].astype(np.float32)
        h = relu(conv3x3(h, w["conv1_w"], w["conv1_b"], True, "f32"))
        h = relu(conv3x3(h, w["conv2_w"], w["conv2_b"], False, mode))
        h = orc._maxpool2(h)
        h = relu(conv3x3(h, w["conv3_w"], w["conv3_b"], True, mode))
        h = relu(conv3x3(h, w["conv4_w"], w["conv4_b"], False, mode))
        h = orc._maxpool2(h)
        h = h.reshape(h.shape[0], -1)
        h = relu(split_dot(h, w["dense1_w"], mode) + w["dense1_b"])
        outs.append(h.dot(w["dense2_w"]) + w["dense2_b"])
    return np.concatenate(outs)


def softmax(z, shift=0.0):
    """scores with the decision boundary moved by `shift` (dense2 bias (+shift/2, -shift/2), tests/test_gpu_cfg4_labels.py)"""
    z = z + np.array([0.5 * shift, -0.5 * shift], np.float32)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=8)
    ap.add_argument("--stride", type=int, default=1, help="every stride-th window")
    args = ap.parse_args()
    N, C = 16000, 128
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, C, 100))
    centers = orc.eval_window_centers(N)[::args.stride]
    xs = []
    t0 = time.time()
    for u in range(args.utterances):
        env = orc.filter_and_envelope(orc.synth_utterance(2028 + u, N), coefs, False)
        xs.append(np.stack([orc.normalize_input(wd) for wd in orc.gather_windows(env, centers)]).astype(np.float32))
    x = np.concatenate(xs)
    print(f"{x.shape[0]} windows of {x.shape[1]} x {x.shape[2]} ({time.time() - t0:.0f} s)", flush=True)
    w = orc.glorot_weights(7)
    z = {"f32": logits(x, w, "f32")}
    check = orc.cnn_forward(x[:256], w)
    print(f"this script's f32 forward vs oracle cnn_forward on 256 windows: max |d score| {np.abs(check - softmax(z['f32'][:256])).max():.2e}")
    for mode in ("3x", "6x"):
        t0 = time.time()
        z[mode] = logits(x, w, mode)
        print(f"{mode} evaluated ({time.time() - t0:.0f} s)", flush=True)
    base = softmax(z["f32"]).astype(np.float64)
    shift = float(np.float32(np.median(np.log(base[:, 1]) - np.log(base[:, 0]))))
    for name, sh in (("glorot7 (BASELINE cfg4: every window scores rising)", 0.0), ("balanced (boundary at the median logit gap)", shift)):
        ref = softmax(z["f32"], sh)
        lab = orc.labels_from_scores(ref)
        margins = np.abs(ref[:, 1] - ref[:, 0])
        print(f"[{name}] {int(lab.sum())} of {len(lab)} rising; windows with an f32 margin below 2e-5: {(margins < 2e-5).sum()}, below 2e-4: {(margins < 2e-4).sum()}")
        for mode in ("3x", "6x"):
            sc = softmax(z[mode], sh)
            differ = np.flatnonzero(orc.labels_from_scores(sc) != lab)
            line = f"[{name}] {mode}: max |score - f32 score| {np.abs(sc - ref).max():.3e}; labels differing from the f32 oracle: {len(differ)} of {len(lab)}"
            if len(differ):
                w64 = dict(w)
                w64["dense2_b"] = np.array([0.5 * sh, -0.5 * sh], np.float32)
                r = orc.cnn_forward_referee(x[differ], w64)
                line += f"; float64 referee margins of those: max {np.abs(r[:, 1] - r[:, 0]).max():.2e}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
