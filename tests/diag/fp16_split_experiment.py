"""Would a split-FP16 CNN keep the labels of cfg4 where the split-bf16 one (tests/diag/bf16_split_experiment.py, the
round-3 default of K4) leaves referee ties?  Numerical experiment on the CPU.

`v_mfma_f32_32x32x16_f16` issues at the rate of the bf16 MFMA. With both GEMM operands split into two fp16 pieces,

    a = a1 + a2,  a1 = fp16(a) , a2 = fp16(a - a1)          (11 + 11 significant bits, against 8 + 8 for bf16)

and the three partial products a1 b1 + a1 b2 + a2 b1 (each exact in float32, float32 accumulation), a product carries
~2^-22 instead of ~2^-16. fp16 has a narrow exponent range (normal from 6.1e-5, 65504 at the top), so the low pieces of
small operands go subnormal unless every layer's operands are scaled by a power of two first ("scaled": activations by
the largest power of two that keeps an upper bound of the layer's input - from the L1 norms of the weights, inputs are
in [0, 1] - below 2^14, weights so that their maximum sits at 2^10 .. 2^11; the product of the two scales is taken out
of the accumulator again, exactly).  conv1 stays float32 here (2 % of the MACs); dense2 + softmax stay float32.

Windows and referee rule as in bf16_split_experiment.py.

Run:  python tests/diag/fp16_split_experiment.py [--utterances 8] [--stride 16]     (CPU only)
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import f2cnn_oracle as orc          # noqa: E402
import bf16_split_experiment as bx  # noqa: E402


def fp16(x):
    """float32 -> nearest float16 (ties to even, subnormals kept, as the hardware conversion), returned as float32"""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def pieces16(x):
    x = np.asarray(x, dtype=np.float32)
    p1 = fp16(x)
    return p1, fp16(x - p1)


def pow2_at_most(v):
    return float(2.0 ** np.floor(np.log2(v)))


def split_dot16(a, b, sa, sb):
    """a (m,k) . b (k,n), operands scaled by the powers of two sa / sb before the split, three partial GEMMs in float32"""
    a1, a2 = pieces16(a * np.float32(sa))
    b1, b2 = pieces16(b * np.float32(sb))
    assert np.isfinite(a1).all() and np.isfinite(b1).all(), "fp16 overflow: scale too large"
    acc = a2.dot(b1)
    acc = acc + a1.dot(b2)
    acc = acc + a1.dot(b1)
    return acc * np.float32(1.0 / (sa * sb))


def layer_scales(w, scaled):
    """per layer (activation scale, weight scale); activation bound from the weights' L1 norms, inputs in [0, 1]"""
    if not scaled:
        return {k: (1.0, 1.0) for k in ("conv2", "conv3", "conv4", "dense1")}, {}
    bound = 1.0
    out, bounds = {}, {}
    bound = float((np.abs(w["conv1_w"]).reshape(-1, w["conv1_w"].shape[-1]).sum(axis=0) * bound + np.abs(w["conv1_b"])).max())
    for name in ("conv2", "conv3", "conv4", "dense1"):
        ww = w[name + "_w"]
        bounds[name] = bound
        sa = pow2_at_most(2.0 ** 14 / bound)
        sb = pow2_at_most(2.0 ** 11 / float(np.abs(ww).max()))
        out[name] = (sa, sb)
        bound = float((np.abs(ww).reshape(-1, ww.shape[-1]).sum(axis=0) * bound + np.abs(w[name + "_b"])).max())
    return out, bounds


def conv3x3(x, w, b, same, scales):
    if same:
        x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    n, H, W, ci = x.shape
    Ho, Wo = H - 2, W - 2
    cols = np.concatenate([x[:, dy:dy + Ho, dx:dx + Wo, :] for dy in range(3) for dx in range(3)], axis=-1)
    out = split_dot16(cols.reshape(-1, 9 * ci), w.reshape(9 * ci, -1), *scales)
    return out.reshape(n, Ho, Wo, -1) + b


def logits(x, w, scales, chunk=512, track=None):
    relu = lambda v: np.maximum(v, 0)
    outs = []
    for s in range(0, x.shape[0], chunk):
        h = x[s:s + chunk, :, :, None].astype(np.float32)
        h = relu(bx.conv3x3(h, w["conv1_w"], w["conv1_b"], True, "f32"))
        for name, same, pool in (("conv2", False, True), ("conv3", True, False), ("conv4", False, True)):
            if track is not None:
                track[name] = max(track.get(name, 0.0), float(h.max()))
            h = relu(conv3x3(h, w[name + "_w"], w[name + "_b"], same, scales[name]))
            if pool:
                h = orc._maxpool2(h)
        h = h.reshape(h.shape[0], -1)
        if track is not None:
            track["dense1"] = max(track.get("dense1", 0.0), float(h.max()))
        h = relu(split_dot16(h, w["dense1_w"], *scales["dense1"]) + w["dense1_b"])
        outs.append(h.dot(w["dense2_w"]) + w["dense2_b"])
    return np.concatenate(outs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=8)
    ap.add_argument("--stride", type=int, default=16, help="every stride-th window")
    ap.add_argument("--with-bf16", action="store_true", help="also evaluate the bf16 3-term split for comparison")
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
    z = {"f32": bx.logits(x, w, "f32")}
    modes = [("fp16 x3, scaled", True), ("fp16 x3, unscaled", False)]
    for name, scaled in modes:
        t0 = time.time()
        scales, bounds = layer_scales(w, scaled)
        track = {}
        z[name] = logits(x, w, scales, track=track)
        print(f"{name} evaluated ({time.time() - t0:.0f} s); " + ", ".join(
            f"{k}: input max {track[k]:.3g}" + (f" (bound {bounds[k]:.3g}, scales 2^{int(np.log2(scales[k][0]))} / 2^{int(np.log2(scales[k][1]))})" if scaled else "")
            for k in track), flush=True)
    if args.with_bf16:
        t0 = time.time()
        z["bf16 x3"] = bx.logits(x, w, "3x")
        print(f"bf16 x3 evaluated ({time.time() - t0:.0f} s)", flush=True)
    base = bx.softmax(z["f32"]).astype(np.float64)
    shift = float(np.float32(np.median(np.log(base[:, 1]) - np.log(base[:, 0]))))
    for wname, sh in (("glorot7 (BASELINE cfg4: every window scores rising)", 0.0), ("balanced (boundary at the median logit gap)", shift)):
        ref = bx.softmax(z["f32"], sh)
        lab = orc.labels_from_scores(ref)
        margins = np.abs(ref[:, 1] - ref[:, 0])
        print(f"[{wname}] {int(lab.sum())} of {len(lab)} rising; windows with an f32 margin below 2e-6: {(margins < 2e-6).sum()}, below 2e-5: {(margins < 2e-5).sum()}")
        for mode in z:
            if mode == "f32":
                continue
            sc = bx.softmax(z[mode], sh)
            differ = np.flatnonzero(orc.labels_from_scores(sc) != lab)
            line = (f"[{wname}] {mode}: max |score - f32 score| {np.abs(sc - ref).max():.3e}, max |logit - f32 logit| "
                    f"{np.abs(z[mode] - z['f32']).max():.3e}; labels differing from the f32 oracle: {len(differ)} of {len(lab)}")
            if len(differ):
                w64 = dict(w)
                w64["dense2_b"] = np.array([0.5 * sh, -0.5 * sh], np.float32)
                r = orc.cnn_forward_referee(x[differ], w64)
                line += f"; float64 referee margins of those: max {np.abs(r[:, 1] - r[:, 0]).max():.2e}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
