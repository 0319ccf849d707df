"""Adversarial search for the accuracy guard of the spectral route (f2cnn_amd/csrc/f2_spectral.hip), on the GPU.

For every length class the spectral kernels serve, N random speech-shaped utterances (tests/speechlike.py: level steps,
bursts, clicks, onsets and offsets, syllable rhythm, voiced harmonics, ramps, DC ...) go through
  (a) the spectral route with its guard switched off (spectral_tol = 1: nothing is handed back) and the guard's own
      per-row values kept (option spectral_guard_dump), and
  (b) the referee: filterbank kernel + float64-FFT envelope kernel (<= 1e-10 of the oracle, tests/test_gpu_envelope.py;
      the worst rows found are re-checked against the CPU oracle here),
for LPF off / 50 / 100 Hz and 64 / 128 channels. Per utterance: true per-channel error max|a - b| / max|b| (the parity bar
of EnvelopeExtraction.py:57-66 output), guard residual / delivered-row maximum, and what the shipped tolerance decides.
Reported: worst error the guard would let through, flagged fraction per signal family, and the seeds of the worst cases.

    python tests/diag/guard_search.py [--per-class 5000] [--out profiles/r05_guard_search.txt] [--classes 13,14,15,16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch                                   # device buffers + reductions only      # noqa: E402
import f2cnn_oracle as orc                     # noqa: E402
import speechlike                              # noqa: E402
from f2cnn_amd import _lib                     # noqa: E402
from f2cnn_amd.gammatone import filters        # noqa: E402

LPFS = [(False, 50.0), (True, 50.0), (True, 100.0)]
CHANNELS = [128, 64]


def class_lengths(log2m, count, rng, min_pad):
    """`count` lengths of the class M = 2^log2m: its extremes (shortest row, least padding the route accepts) + random ones"""
    lo, hi = (1 << (log2m - 1)) + 1, (1 << log2m) - min_pad
    fixed = [lo, lo + 1, hi, hi - 1, (lo + hi) // 2]
    if log2m == 14:
        fixed += [16000] * 6                    # the benchmark's length
    out = fixed[:count]
    while len(out) < count:
        out.append(int(rng.integers(lo, hi + 1)))
    return out


def run_config(ctx, d_wave, offs, coefs, B, C, n, lpf, cutoff, env_s, env_r, min_pad):
    """-> per-row arrays (B, C): true error of the spectral route, guard values (gi, go, glp)"""
    with ctx.options(spectral=1, spectral_min_rows=0, spectral_tol=1.0, spectral_guard_dump=1, spectral_min_pad=min_pad):
        ctx.filterbank_envelope_fused(d_wave.data_ptr(), _lib.WAVE_I16, offs, coefs, B, C, lpf, cutoff, _lib.FFT_F32,
                                      env_s.data_ptr(), None, _lib.MEM_DEVICE)
        routed = int(ctx.get_option("spectral_routed"))
        g = ctx.spectral_guard_values().reshape(B, C, 4)
    assert routed == B, (routed, B, n)
    with ctx.options(spectral=0):
        ctx.filterbank_envelope_fused(d_wave.data_ptr(), _lib.WAVE_I16, offs, coefs, B, C, lpf, cutoff, _lib.FFT_F64,
                                      env_r.data_ptr(), None, _lib.MEM_DEVICE)
    ctx.synchronize()
    a = env_s[:B * C * n].view(B, C, n)
    b = env_r[:B * C * n].view(B, C, n)
    assert not torch.isnan(a).any()
    scale = b.abs().amax(dim=2)
    err = (a - b).abs().amax(dim=2)
    rel = torch.where(scale > 0, err / scale.clamp_min(1e-300), torch.zeros_like(err))
    zero_ok = bool((err[scale == 0] == 0).all())
    torch.cuda.synchronize()
    return rel.cpu().numpy(), g, zero_ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-class", type=int, default=5000)
    ap.add_argument("--classes", default="13,14,15,16")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tol", type=float, default=None, help="tolerance to evaluate (default: the library's)")
    ap.add_argument("--npz", default=None, help="per-utterance arrays for offline analysis")
    ap.add_argument("--min-pad", type=int, default=-1, help="padding samples a row needs for the route (-1: the library's rule, "
                    "256 for this bank; 64 shows what the rule is for)")
    ap.add_argument("--low-freq", type=float, default=100.0, help="lowest centre frequency of the bank (the reference's config: 100)")
    ap.add_argument("--rows", action="store_true", help="... and the per-row arrays of the 128-channel configurations")
    args = ap.parse_args()
    ctx = _lib.Context(0)
    tol = args.tol if args.tol is not None else ctx.get_option("spectral_tol")
    lines = []

    def say(s=""):
        print(s, flush=True)
        lines.append(s)

    say(f"# guard search: {args.per_class} utterances per length class, spectral_tol = {tol:g}, spectral_min_pad = {args.min_pad}, low_freq = {args.low_freq:g}, families = {', '.join(speechlike.FAMILIES)}")
    coefs = {C: filters.make_erb_filters(16000, filters.centre_freqs(16000, C, args.low_freq)) for C in CHANNELS}
    dump = {}
    overall_worst_unflagged = 0.0
    for log2m in [int(x) for x in args.classes.split(",")]:
        t_class = time.time()
        rng = np.random.default_rng(log2m)
        if args.min_pad < 0:      # the library's rule: 1.1 x the ringing peak time of the slowest channel, rounded up to 32
            a2 = coefs[CHANNELS[0]][:, 8] / coefs[CHANNELS[0]][:, 6]
            pad = (int(np.ceil(1.1 * (3.0 / (-0.5 * np.log(a2))).max())) + 31) // 32 * 32
        else:
            pad = args.min_pad
        nmax = (1 << log2m) - pad
        Bb = max(8, min(250, (1 << 31) // (8 * 128 * nmax)))
        nbatch = (args.per_class + Bb - 1) // Bb
        lens = class_lengths(log2m, nbatch, rng, pad)
        env_s = torch.empty(Bb * 128 * nmax, dtype=torch.float64, device="cuda")
        env_r = torch.empty_like(env_s)
        meta = []                                              # (seed, n, family, par)
        per = {(C, lpf, cut): {"err": [], "g": []} for C in CHANNELS for lpf, cut in LPFS}
        seed = 1000 * log2m
        for bi, n in enumerate(lens):
            B = min(Bb, args.per_class - bi * Bb)
            waves = []
            for _ in range(B):
                w, fam, par = speechlike.make(seed, n)
                meta.append((seed, n, fam, par))
                waves.append(w)
                seed += 1
            d_wave = torch.from_numpy(np.concatenate(waves)).cuda()
            offs = np.arange(B + 1, dtype=np.int64) * n
            for C in CHANNELS:
                for lpf, cut in LPFS:
                    rel, g, zero_ok = run_config(ctx, d_wave, offs, coefs[C], B, C, n, lpf, cut, env_s, env_r, args.min_pad)
                    assert zero_ok, "an all-zero reference row came out non-zero"
                    per[(C, lpf, cut)]["err"].append(rel)
                    per[(C, lpf, cut)]["g"].append(g)
            if bi % 10 == 0:
                print(f"  [class 2^{log2m}] batch {bi + 1}/{nbatch} (n = {n}) {time.time() - t_class:.0f} s", file=sys.stderr, flush=True)
        del env_s, env_r
        torch.cuda.empty_cache()
        fam = np.array([m[2] for m in meta])
        say(f"\n## class M = 2^{log2m} (rows of {(1 << (log2m - 1)) + 1}..{nmax} samples), {len(meta)} utterances, {nbatch} lengths, {time.time() - t_class:.0f} s")
        for (C, lpf, cut), d in per.items():
            err = np.concatenate(d["err"])                     # (N, C)
            g = np.concatenate(d["g"])
            den = g[..., 2] if lpf else g[..., 0]
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = np.where(den > 0, g[..., 1] / den, np.where(g[..., 1] > 0, np.inf, 0.0))
                ratio_old = np.where(g[..., 0] > 0, g[..., 1] / g[..., 0], np.where(g[..., 1] > 0, np.inf, 0.0))
            u_err, u_ratio, u_old = err.max(axis=1), ratio.max(axis=1), ratio_old.max(axis=1)
            flagged = u_ratio > tol
            flagged_old = u_old > tol
            wu = float(u_err[~flagged].max()) if (~flagged).any() else 0.0
            wu_old = float(u_err[~flagged_old].max()) if (~flagged_old).any() else 0.0
            overall_worst_unflagged = max(overall_worst_unflagged, wu)
            # smallest tolerance-independent fact: the largest error among utterances with guard ratio below r, as a curve
            order = np.argsort(u_ratio)
            cum = np.maximum.accumulate(u_err[order])
            safe = u_ratio[order][np.searchsorted(cum, 5e-6, side="right") - 1] if (cum <= 5e-6).any() else 0.0
            name = f"C = {C:3d}, " + (f"LPF {cut:g} Hz" if lpf else "no LPF   ")
            say(f"{name}: worst error with the guard off {u_err.max():.2e}; tol {tol:g} flags {int(flagged.sum())} "
                f"({100.0 * flagged.mean():.2f} %), worst unflagged {wu:.2e}   [round-4 rule (raw maximum): flags "
                f"{int(flagged_old.sum())}, worst unflagged {wu_old:.2e}]; largest tol with worst unflagged <= 5e-6: {safe:.2e}")
            if C == 128:
                parts = []
                for f in speechlike.FAMILIES:
                    sel = fam == f
                    parts.append(f"{f} {int(flagged[sel].sum())}/{int(sel.sum())}")
                say("    flagged per family: " + ", ".join(parts))
                top = np.argsort(np.where(flagged, -1.0, u_err))[::-1][:5]
                for i in top:
                    c = int(err[i].argmax())
                    say(f"    unflagged worst: seed {meta[i][0]} n {meta[i][1]} {meta[i][2]} {json.dumps(meta[i][3])} channel {c} "
                        f"error {u_err[i]:.2e} guard ratio {u_ratio[i]:.2e}")
                # referee check against the CPU oracle on the single worst unflagged utterance
                i = int(top[0])
                w, _, _ = speechlike.make(meta[i][0], meta[i][1], meta[i][2])
                ref = orc.filter_and_envelope(w, coefs[C], lpf, cut)
                out = np.empty(C * len(w))
                with ctx.options(spectral=0):
                    ctx.filterbank_envelope_fused(w, _lib.WAVE_I16, np.array([0, len(w)], np.int64), coefs[C], 1, C, lpf, cut,
                                                  _lib.FFT_F64, out, None, _lib.MEM_HOST)
                scale = np.abs(ref).max(axis=1)
                rr = (np.abs(out.reshape(C, -1) - ref).max(axis=1)[scale > 0] / scale[scale > 0]).max()
                say(f"    referee vs CPU oracle on that utterance: {rr:.1e}")
            if args.rows and C == 128:
                dump[f"c{log2m}_{C}_{int(lpf) * int(cut)}_rows_err"] = err.astype(np.float32)
                dump[f"c{log2m}_{C}_{int(lpf) * int(cut)}_rows_g"] = g.astype(np.float32)
            dump[f"c{log2m}_{C}_{int(lpf) * int(cut)}_err"] = u_err.astype(np.float32)
            dump[f"c{log2m}_{C}_{int(lpf) * int(cut)}_ratio"] = u_ratio.astype(np.float32)
            dump[f"c{log2m}_{C}_{int(lpf) * int(cut)}_ratio_old"] = u_old.astype(np.float32)
        dump[f"c{log2m}_family"] = fam
        dump[f"c{log2m}_seed_n"] = np.array([(m[0], m[1]) for m in meta], np.int64)
    say(f"\n# worst unflagged error over everything: {overall_worst_unflagged:.2e} (bar 1e-5, target <= 5e-6)")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")
    if args.npz:
        np.savez_compressed(args.npz, **dump)
    ctx.close()


if __name__ == "__main__":
    main()
