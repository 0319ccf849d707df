#!/usr/bin/env python3
"""
bench.py -- throughput of the F2CNN hot path on MI355X, in audio-seconds per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4] [--batch B]

One "step" = one pass of the hot path over one batch of synthetic utterances that is already
resident in HBM (int16 waves in, float64 envelopes out, all device pointers through the C ABI).
Workloads (BASELINE.json configs):
  cfg3 (default)  fused filterbank + Hilbert envelope + 50 Hz LPF, 1000 x 1 s utterances, 128 channels
  cfg2            128-channel filterbank only, 256 utterances
  cfg4            `cnn eval` end to end (filterbank, envelope, every-sample windows, CNN), 8 utterances
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank processes its own batch
(utterances shard with no collective: weak scaling); the only cross-rank traffic is the barrier and
the MAX-reduction of the elapsed time. Rank 0 prints one JSON line.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 16000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F32_PEAK_TFLOPS = 157.3        # f32 vector/matrix peak, used for the CNN-bound workload


def synth_batch(seed, first, count, n):
    """Utterance u of the corpus: default_rng(seed+u) Gaussian noise, sigma 3000, int16."""
    out = np.empty((count, n), dtype=np.int16)
    for i in range(count):
        rng = np.random.default_rng(seed + first + i)
        out[i] = np.clip(np.round(rng.standard_normal(n) * 3000.0), -32768, 32767).astype(np.int16)
    return out


def _cpu_one(args):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import f2cnn_oracle as orc
    seed, idx, n, C, lpf, mode = args
    wave = synth_batch(seed, idx, 1, n)[0]
    coefs = orc.make_erb_filters(FS, orc.centre_freqs(FS, C, 100))
    t = time.perf_counter()
    if mode == "filterbank":
        orc.erb_filterbank(wave, coefs)
    else:
        orc.filter_and_envelope(wave, coefs, bool(lpf), lpf or 100)
    return time.perf_counter() - t


def cpu_baseline(seed, n, C, lpf, mode, sample):
    """The oracle (SciPy lfilter/hilbert restatement of the reference) on the host cores, compute only."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)   # the one-GPU box's CPU share
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_one, [(seed, i, 256, 8, lpf, mode) for i in range(cores)])  # start + import the workers
        t0 = time.perf_counter()
        per = pool.map(_cpu_one, [(seed, i, n, C, lpf, mode) for i in range(sample)], chunksize=1)
        wall = time.perf_counter() - t0
    return {"value": round(sample * n / FS / wall, 3), "unit": "audio-seconds/s", "cores": cores, "kind": "port",
            "sample": f"{sample} of the same synthetic utterances ({n} samples, {C} channels), oracle/f2cnn_oracle.py "
                      f"{'erb_filterbank' if mode == 'filterbank' else 'filter_and_envelope'} in a {cores}-process pool, "
                      f"compute only; {sum(per):.1f} s of CPU work in {wall:.1f} s wall",
            "single_core_value": round(sample * n / FS / sum(per), 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step (default: the config's)")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--samples", type=int, default=FS, help="samples per utterance")
    ap.add_argument("--fft", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world

    import torch
    import torch.distributed as dist
    from f2cnn_amd import _lib
    from f2cnn_amd.gammatone import filters

    # one rank per GPU over RCCL; F2CNN_BENCH_BACKEND=gloo + F2CNN_BENCH_ONE_DEVICE=1 rehearse the multi-rank
    # path on a single GPU (all ranks on device 0, CPU tensors for the barrier / MAX-reduce)
    backend = os.environ.get("F2CNN_BENCH_BACKEND", "nccl")
    if os.environ.get("F2CNN_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ctx = _lib.Context(local)
    C, N = args.channels, args.samples
    coefs = filters.make_erb_filters(FS, filters.centre_freqs(FS, C, 100))
    precision = _lib.FFT_F32 if args.fft == "f32" else _lib.FFT_F64
    seeds = {"cfg2": 2026, "cfg3": 2027, "cfg4": 2028}
    B = args.batch or {"cfg2": 256, "cfg3": 1000, "cfg4": 8}[args.workload]
    seed = seeds[args.workload]
    lpf = 50 if args.workload == "cfg3" else 0
    if args.workload == "cfg4" and args.steps == 20 and args.warmup == 3:
        args.steps, args.warmup = 3, 1

    waves = synth_batch(seed, rank * B, B, N)
    offsets = np.arange(B + 1, dtype=np.int64) * N
    d_wave = ctx.malloc(waves.nbytes)
    ctx.h2d(d_wave, waves)
    d_out = ctx.malloc(8 * C * N * B)

    CNN_FLOP_PER_WINDOW = 2 * 20990472          # SURVEY 8a row a13 (11 x 128 window)
    bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
    if args.workload == "cfg2":
        def step():
            ctx.erb_filterbank_batch(d_wave, _lib.WAVE_I16, offsets, coefs, B, C, d_out, _lib.MEM_DEVICE)
        algo = {"k_erb_filterbank": B * (2 * N + 8 * C * N)}
        label = f"cfg2: {C}-channel gammatone filterbank, batch of {B} x {N / FS:g} s utterances per GPU"
    elif args.workload == "cfg3":
        def step():
            ctx.filterbank_envelope_fused(d_wave, _lib.WAVE_I16, offsets, coefs, B, C, True, lpf, precision, d_out,
                                          None, _lib.MEM_DEVICE)
        # float32 FFT: the filterbank hands its rows to the envelope kernel as float32 inside the ENV buffer
        # (4 B written + 4 B read per sample-channel); float64 FFT: float64 hand-off (8 + 8)
        h = 4 if args.fft == "f32" and N <= 16384 else 8
        algo = {"k_erb_filterbank": B * (2 * N + h * C * N), "k_envelope": B * (h + 8) * C * N}
        label = (f"cfg3: fused filterbank + Hilbert envelope + {lpf} Hz LPF, batch of {B} x {N / FS:g} s utterances "
                 f"per GPU, {C} channels, ENV1 (float64) out")
    else:
        from f2cnn_amd.model import F2CNNModel
        if C != 128:
            raise SystemExit("cfg4 uses the 11 x 128 network")
        model = F2CNNModel.glorot(7)
        hcnn = model.handle(ctx)
        nb = N - 11 * 160
        d_scores = ctx.malloc(8 * nb * B)
        d_labels = ctx.malloc(nb * B)

        def step():
            ctx.eval_batch(hcnn, d_wave, _lib.WAVE_I16, offsets, coefs, B, C, False, 0.0, precision, 5, 160, d_scores,
                           d_labels, _lib.MEM_DEVICE)
        # flop per CNN launch group: filled in from the measured number of launches (the library chunks the windows)
        algo = {"k_cnn_forward": None, "k_gather_windows": 0,
                "k_erb_filterbank": 2 * N + 8 * C * N, "k_envelope": 16 * C * N}
        bound, peak, unit = "mfma", F32_PEAK_TFLOPS * 1e3, "GFLOP/s"
        label = (f"cfg4: cnn eval end to end (filterbank, envelope, every-sample 11x{C} windows, normalise, CNN), "
                 f"{B} x {N / FS:g} s utterances per GPU = {B * nb} windows, Glorot weights seed 7")

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    if args.workload == "cfg4" and "k_cnn_forward" in prof:
        algo["k_cnn_forward"] = CNN_FLOP_PER_WINDOW * nb * B * args.steps / prof["k_cnn_forward"][0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    if rank == 0:
        audio_s = world * B * N / FS * args.steps
        # dominant kernel = largest summed device time (HIP events around every launch, on the launch stream)
        kname, (launches, total_ms) = max(prof.items(), key=lambda kv: kv[1][1])
        avg_s = total_ms / launches / 1e3
        achieved = algo[kname] / avg_s / 1e9
        pipeline_bytes = B * (2 * N + 8 * C * N)     # what the whole step must move at least (cfg2/cfg3)
        # HBM bytes per launch from the PMC counters (collected separately with rocprofv3 --pmc and committed
        # under profiles/; only valid for the batch it was measured on)
        traffic = None
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_cfg3.json")))   # latest measurement
        tpath = tfiles[-1] if tfiles else ""
        if args.workload == "cfg3" and tpath:
            tj = json.load(open(tpath))
            if tj.get("batch") == B and kname in tj and C == 128 and N == FS and args.fft == "f32":
                traffic = tj[kname]["hbm_bytes_per_launch"]
        out = {
            "metric": "audio-seconds/sec through filterbank+envelope" + ("+CNN" if args.workload == "cfg4" else "")
                      + " (HIP, 1 MI355X per rank)",
            "value": round(audio_s / elapsed, 1),
            "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64" if args.workload == "cfg2" else ("f64 IIR + %s FFT" % args.fft) + (
                " + f32 CNN" if args.workload == "cfg4" else ""),
            "data": "synthetic",
            "config": {"workload": label, "batch_per_gpu": B, "channels": C, "samples_per_utterance": N,
                       "sample_rate": FS, "lpf_hz": lpf, "parallelism": f"utterance-sharded x{world}, no collective"},
            "roofline": {"bound": bound, "kernel": kname, "achieved": round(achieved, 1), "peak": peak,
                         "unit": unit, "frac": round(achieved / peak, 4), "traffic": traffic,
                         ("algorithmic_bytes_per_launch" if bound == "hbm" else "algorithmic_flop_per_launch"):
                             algo[kname], "avg_launch_ms": round(avg_s * 1e3, 4), "launches_timed": launches},
            "kernels": {k: {"launches": n, "avg_ms": round(ms / n, 4),
                            ("algorithmic_GBps" if k != "k_cnn_forward" else "algorithmic_GFLOPps"):
                                round(algo[k] / (ms / n / 1e3) / 1e9, 1)} for k, (n, ms) in prof.items()},
        }
        if args.workload != "cfg4":
            out["pipeline"] = {"algorithmic_bytes_per_step": pipeline_bytes,
                               "algorithmic_GBps": round(pipeline_bytes / (elapsed / args.steps) / 1e9, 1),
                               "frac_of_hbm_peak": round(pipeline_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}
        if not args.no_cpu_baseline:
            if args.workload == "cfg4":
                out["cpu_baseline"] = None   # the CNN oracle (NumPy) is timed by tests only; no Keras on the box
            else:
                out["cpu_baseline"] = cpu_baseline(seed, N, C, lpf, "filterbank" if args.workload == "cfg2" else "both",
                                                   args.cpu_sample)
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)

    ctx.free(d_wave)
    ctx.free(d_out)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
