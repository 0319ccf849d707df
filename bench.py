#!/usr/bin/env python3
"""
bench.py -- throughput of the F2CNN hot path on MI355X, in audio-seconds per second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg5|cfg5r|cfg2|cfg4] [--batch B]

One "step" = one pass of the hot path over synthetic utterances that are already resident in HBM (int16 waves in,
float64 envelopes out, device pointers through the C ABI). Workloads (BASELINE.json configs):
  cfg3   fused filterbank + Hilbert envelope + 50 Hz LPF, 1000 x 1 s utterances per GPU, 128 channels   (weak scaling)
  cfg5   the 10 000-utterance corpus (utterance u = seed 2029+u), rank r owns utterances r::G, processed in
         HBM-sized batches through the same fused call; a step = one pass over the rank's shard            (strong scaling)
  cfg5r  the same corpus with lengths U[16000, 64000] samples (SURVEY 8d secondary run)
  cfg2   128-channel filterbank only, 256 utterances
  cfg4   `cnn eval` end to end (filterbank, envelope, every-sample windows, CNN), 8 utterances
Default workload: cfg3 at --gpus 1 (plus side blocks: float64-FFT timing, cfg4, cfg5, end-to-end files), cfg5 at
--gpus N > 1.

Multi-GPU: one process per GPU, no data-path collective (utterances are independent; reference analogue: the
Pool(cpu_count()) over files at scripts/processing/GammatoneFiltering.py:121-125). `python bench.py --gpus N` starts
the N rank processes itself (fresh children, before anything here touches the GPU); under torch.distributed.run the
ranks already exist. Either way the ranks meet in a gloo (CPU) process group for the start barrier and the gather of
per-rank results; rank 0 merges them on the host (sum of audio-seconds / MAX elapsed) and prints ONE JSON line.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 16000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is what a copy kernel reaches)
F32_PEAK_TFLOPS = 157.3        # f32 vector/matrix peak, used for the CNN-bound workload
F64_PEAK_TFLOPS = 78.6         # f64 vector peak (spec); tools/ubench/fma64_operands.hip reaches 55-62 with 2-8 waves/SIMD
K1_FLOP_PER_SAMPLE_CHANNEL = 2 * 13   # 13 float64 FMAs per sample-channel
CNN_FLOP_PER_WINDOW = 2 * 20990472    # SURVEY 8a row a13 (11 x 128 window)
CNN_SPLIT_FLOP_PER_WINDOW = 2 * 9 * (32 * 32 * 9 * 126 + 32 * 64 * 4 * 63 + 64 * 64 * 2 * 61) + 2 * 1920 * 516   # conv2-conv4 + dense1
BF16_PEAK_TFLOPS = 2500.0      # dense 16-bit (fp16 = bf16) matrix peak (MI355X_MICROARCH.md)


def cnn_accounting(ctx, flop, cnn_s, windows):
    """The CNN stage against the matrix peak of the arithmetic it runs in. With option cnn_f16x3 (default) conv2-conv4 and
    dense1 issue three v_mfma_f32_32x32x16_f16 per product (operands split in two fp16 pieces, f32 accumulation): `issued`
    flops = 3 x those layers' flops + the rest at face value, priced against the fp16 (= bf16) peak; TFLOPps stays the algorithmic
    (float32-equivalent) rate, which is what audio-seconds/s follows."""
    split = bool(ctx.get_option("cnn_f16x3"))
    out = {"TFLOPps": round(flop / cnn_s / 1e12, 2), "f32_matrix_peak_TFLOPps": F32_PEAK_TFLOPS,
           "vs_f32_matrix_peak": round(flop / cnn_s / 1e12 / F32_PEAK_TFLOPS, 4), "flop_per_window": CNN_FLOP_PER_WINDOW}
    if split:
        issued = flop + 2 * CNN_SPLIT_FLOP_PER_WINDOW * windows
        out.update({"arithmetic": "conv2-conv4, dense1: v_mfma_f32_32x32x16_f16 x 3 per product (operands scaled by powers of two "
                                  "and split in two fp16 pieces, f32 accumulate: float32 rounding level); conv1 likewise on the "
                                  "weight-stationary path; dense2: f32",
                    "issued_TFLOPps": round(issued / cnn_s / 1e12, 1), "peak_TFLOPps": BF16_PEAK_TFLOPS,
                    "frac": round(issued / cnn_s / 1e12 / BF16_PEAK_TFLOPS, 4),
                    "frac_is": "ISSUED flops (three MFMAs per product) / fp16 (= bf16) dense peak",
                    "algorithmic_frac_of_bf16_peak": round(flop / cnn_s / 1e12 / BF16_PEAK_TFLOPS, 4),
                    "kernels": "weight-stationary (cnn_ws)" if ctx.get_option("cnn_ws") else "one workgroup per tile"})
    else:
        out.update({"arithmetic": "v_mfma_f32_32x32x2_f32 throughout", "peak_TFLOPps": F32_PEAK_TFLOPS,
                    "frac": out["vs_f32_matrix_peak"]})
    return out
CORPUS = 10000
SEEDS = {"cfg2": 2026, "cfg3": 2027, "cfg4": 2028, "cfg5": 2029, "cfg5r": 2029}


# ------------------------------------------------------------------------------------------------
# synthetic corpus
# ------------------------------------------------------------------------------------------------
def synth_utterance(seed, n):
    rng = np.random.default_rng(seed)
    return np.clip(np.round(rng.standard_normal(n) * 3000.0), -32768, 32767).astype(np.int16)


def synth_batch(seed, first, count, n):
    """Utterances first..first+count-1 of a corpus: utterance u = default_rng(seed+u) Gaussian noise, sigma 3000."""
    out = np.empty((count, n), dtype=np.int16)
    for i in range(count):
        out[i] = synth_utterance(seed + first + i, n)
    return out


def ragged_lengths(seed=2029, count=CORPUS):
    """Utterance lengths of the variable-length variant of cfg5: U[16000, 64000] samples."""
    return np.random.default_rng(seed).integers(16000, 64001, size=count).astype(np.int64)


def _synth_many(job):
    seed, idx, lens = job
    return np.concatenate([synth_utterance(seed + int(u), int(n)) for u, n in zip(idx, lens)])


def synth_corpus(seed, idx, lens, procs):
    """Concatenated waves of the utterances `idx` (lengths `lens`), generated by a process pool."""
    import multiprocessing as mp
    if len(idx) == 0:
        return np.zeros(0, np.int16)
    chunks = max(1, min(len(idx), procs * 4))
    jobs = [(seed, a, b) for a, b in zip(np.array_split(idx, chunks), np.array_split(lens, chunks))]
    if procs <= 1:
        return np.concatenate([_synth_many(j) for j in jobs])
    pool = mp.get_context("spawn").Pool(procs)
    try:
        parts = pool.map(_synth_many, jobs)
    finally:
        pool.close()      # (Pool.__exit__ would terminate() the workers: SIGTERM noise in a profiler's log)
        pool.join()
    return np.concatenate(parts)


def cgroup_cpu_quota():
    """CPUs the container's cgroup lets this process use at once (cpu.max / cfs quota), or None if unlimited / unknown"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else max(1, quota // period)
    except (OSError, ValueError):
        return None


def host_cores():
    """Worker processes of the CPU baselines: every core this process may run on (its affinity mask - after the rank pinned
    itself to its GPU's NUMA node - capped by the cgroup's CPU quota); $F2CNN_BENCH_CORES overrides. (Rounds 1-4 capped
    this at 16; BASELINE.md section 3 plans P = os.cpu_count().)"""
    if os.environ.get("F2CNN_BENCH_CORES"):
        return max(1, int(os.environ["F2CNN_BENCH_CORES"]))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_cpu_quota()
    return max(1, min(cores, quota) if quota else cores)


def host_core_facts(used):
    """What the host offers beside what a baseline used, so that a reader can rescale: `cores` in a cpu_baseline is `used`."""
    return {"cores_used": used, "cores_in_affinity_mask": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cores_of_the_host": os.cpu_count(), "cgroup_cpu_quota": cgroup_cpu_quota()}


def all_cores_estimate(gpu_value, cpu):
    """GPU / CPU ratio if the baseline had used every core of the host instead of `cores` (linear scaling assumed)"""
    total = os.cpu_count() or cpu["cores"]
    return round(gpu_value / (cpu["value"] * total / cpu["cores"]), 1)


# ------------------------------------------------------------------------------------------------
# CPU baselines: the oracle (restatement of the reference, oracle/f2cnn_oracle.py) on the host cores
# ------------------------------------------------------------------------------------------------
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
    """Filterbank (+ envelope) of `sample` utterances of the workload, one utterance per pool task, compute only."""
    import multiprocessing as mp
    cores = host_cores()
    sample = max(sample, 2 * cores)       # two utterances per worker at least: the pool's tail does not dominate
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_cpu_one, [(seed, i, 256, 8, lpf, mode) for i in range(cores)])  # start + import the workers
        t0 = time.perf_counter()
        per = pool.map(_cpu_one, [(seed, i, n, C, lpf, mode) for i in range(sample)], chunksize=1)
        wall = time.perf_counter() - t0
    return {"value": round(sample * n / FS / wall, 3), "unit": "audio-seconds/s", "cores": cores, "kind": "port",
            "host": host_core_facts(cores),
            "sample": f"{sample} of the same synthetic utterances ({n} samples, {C} channels), oracle/f2cnn_oracle.py "
                      f"{'erb_filterbank' if mode == 'filterbank' else 'filter_and_envelope'} in a {cores}-process pool, "
                      f"compute only; {sum(per):.1f} s of CPU work in {wall:.1f} s wall",
            "single_core_value": round(sample * n / FS / sum(per), 3)}


def _cpu_cnn_one(args):
    """One pool task of the cfg4 baseline: `count` consecutive every-sample windows of utterance 0, DSP included once."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import f2cnn_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)   # one BLAS thread per pool process (the pool owns the cores)
    except ImportError:
        limit = None
    seed, first, count, n = args
    wave = synth_batch(seed, 0, 1, n)[0]
    coefs = orc.make_erb_filters(FS, orc.centre_freqs(FS, 128, 100))
    weights = orc.glorot_weights(7)
    t0 = time.perf_counter()
    env = orc.filter_and_envelope(wave, coefs, False, 100)
    t1 = time.perf_counter()
    centers = orc.eval_window_centers(n)[first:first + count]
    w = orc.gather_windows(env, centers)
    for i in range(w.shape[0]):
        w[i] = orc.normalize_input(w[i])
    t2 = time.perf_counter()
    orc.cnn_forward(w, weights)
    t3 = time.perf_counter()
    del limit
    return t1 - t0, t2 - t1, t3 - t2


def cpu_baseline_cnn(seed, n, windows_per_task=256):
    """`cnn eval` on the host cores: per audio-second = DSP of one utterance + (n - 1760) windows gathered,
    normalised and run through the CNN oracle. Timed on a bounded sample (one task per core) and scaled."""
    import multiprocessing as mp
    cores = host_cores()
    nb = n - 11 * 160
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_cpu_cnn_one, [(seed, 0, 2, 2000)] * cores)
        t0 = time.perf_counter()
        per = pool.map(_cpu_cnn_one, [(seed, (i * windows_per_task) % max(nb - windows_per_task, 1), windows_per_task, n)
                                      for i in range(cores)], chunksize=1)
        wall = time.perf_counter() - t0
    dsp = float(np.mean([p[0] for p in per]))
    win = float(np.mean([p[1] + p[2] for p in per])) / windows_per_task       # s per window on one core
    per_utt_core_s = dsp + win * nb
    return {"value": round(cores * (n / FS) / per_utt_core_s, 4), "unit": "audio-seconds/s", "cores": cores, "kind": "port",
            "host": host_core_facts(cores),
            "sample": f"{cores} pool tasks x {windows_per_task} every-sample windows of utterance 0 (filterbank + envelope "
                      f"{dsp:.2f} s, gather + normalise + oracle cnn_forward {win * 1e3:.2f} ms per window on one core, one "
                      f"BLAS thread per process), scaled to {nb} windows per utterance and {cores} cores; {wall:.1f} s wall",
            "cnn_windows_per_s_per_core": round(1.0 / win, 1)}


# ------------------------------------------------------------------------------------------------
# multi-rank plumbing
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has
    not imported torch or touched HIP and never does), pass their output through, return the worst exit status."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                for q in live:      # a rank died: the others would wait for it at the next barrier
                    q.terminate()
    return rc


class Ranks:
    """Start barrier + gather of per-rank results over a gloo (CPU) process group; trivial for one rank."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.dist = None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def gather(self, obj):
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def lib_source_hash():
    """Hash of the kernel sources: a PMC traffic file is only quoted for the build it was measured on."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "f2cnn_amd", "csrc", "*"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(workload, B, C, N, fft):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary of this build (profiles/), or None."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_cfg3.json")))
    if workload != "cfg3" or not files:
        return None
    tj = json.load(open(files[-1]))
    if tj.get("batch") != B or C != 128 or N != FS or fft != "f32" or tj.get("lib_source_hash") != lib_source_hash():
        return None
    return {k: v["hbm_bytes_per_launch"] for k, v in tj.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}


class DspJob:
    """Filterbank (+ envelope) over a list of batches resident in HBM. One step = one call per batch."""

    def __init__(self, ctx, coefs, C, waves, lens, batch, mode, lpf, precision, out_budget_bytes=200 << 30):
        from f2cnn_amd import _lib
        self.ctx, self.coefs, self.C, self.mode, self.lpf, self.precision = ctx, coefs, C, mode, lpf, precision
        self._lib = _lib
        lens = np.asarray(lens, dtype=np.int64)
        self.total_samples = int(lens.sum())
        self.d_wave = ctx.malloc(max(waves.nbytes, 2))
        ctx.h2d(self.d_wave, waves)
        nb = max(1, -(-len(lens) // batch))
        self.batches = []
        start = np.concatenate([[0], np.cumsum(lens)])
        bounds = np.linspace(0, len(lens), nb + 1).astype(int)     # equal-sized batches
        for a, b in zip(bounds[:-1], bounds[1:]):
            if b > a:
                self.batches.append((int(start[a]), np.concatenate([[0], np.cumsum(lens[a:b])]).astype(np.int64), b - a))
        out_bytes = [8 * C * int(off[-1]) for _, off, _ in self.batches]
        # every batch keeps its own output region while the shard fits the budget (288 GB of HBM: the whole 10 000
        # x 1 s corpus = 164 GB of envelopes stays resident); otherwise the batches share one region
        self.resident = sum(out_bytes) <= out_budget_bytes
        self.d_out = ctx.malloc(sum(out_bytes) if self.resident else max(out_bytes))
        pos, self.out_ptr = 0, []
        for ob in out_bytes:
            self.out_ptr.append(self.d_out + pos)
            pos += ob if self.resident else 0

    def step(self):
        L = self._lib
        for (w0, off, nb), dst in zip(self.batches, self.out_ptr):
            if self.mode == "filterbank":
                self.ctx.erb_filterbank_batch(self.d_wave + 2 * w0, L.WAVE_I16, off, self.coefs, nb, self.C, dst, L.MEM_DEVICE)
            else:
                self.ctx.filterbank_envelope_fused(self.d_wave + 2 * w0, L.WAVE_I16, off, self.coefs, nb, self.C,
                                                   bool(self.lpf), self.lpf or 100, self.precision, dst, None, L.MEM_DEVICE)

    def routing(self):
        """Samples per step that went through the spectral kernel, and those of them its accuracy guard sent back to the
        filterbank kernel + envelope kernel: one extra, untimed pass that asks the library after every launch."""
        L = self._lib
        routed = flagged = 0
        if self.mode == "filterbank":
            return 0, 0
        for (w0, off, nb), dst in zip(self.batches, self.out_ptr):
            self.ctx.filterbank_envelope_fused(self.d_wave + 2 * w0, L.WAVE_I16, off, self.coefs, nb, self.C,
                                               bool(self.lpf), self.lpf or 100, self.precision, dst, None, L.MEM_DEVICE)
            routed += int(self.ctx.get_option("spectral_routed_samples"))
            flagged += int(self.ctx.get_option("spectral_flagged_samples"))
        return routed, flagged

    def free(self):
        self.ctx.free(self.d_wave)
        self.ctx.free(self.d_out)


def timed(ctx, ranks, step, steps, warmup):
    """W untimed steps, barrier + device sync, K timed steps, device sync. Returns (elapsed_s, prof): the library brackets every
    kernel group with HIP events inside the timed region (what `roofline.achieved` is made of; measured on cfg4, 34 event pairs
    per step: a pass without them is no faster)."""
    for _ in range(warmup):
        step()
    ctx.synchronize()
    ranks.barrier()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    return elapsed, prof


def dsp_kernel_report(prof, steps, C, samples_per_step, mode, handoff_bytes, routed=0, flagged=0):
    """Per-kernel figures for one rank. Required (algorithmic) bytes follow SURVEY 8d: the step must read 2 B per
    sample and write 8*C B per sample; the K1->K2 hand-off is overhead, not algorithm. K1 is bound by the float64
    FMA pipe, so it also gets a flop/s figure against the f64 peak. Every kernel is charged the samples IT processed:
    `routed` samples per step go through the spectral kernels, the filterbank kernel + envelope kernel see the rest plus the
    `flagged` ones the accuracy guard sent back - launches that only walk the flags are reported as such, with no rate."""
    own = {"k_erb_filterbank": samples_per_step - routed + flagged, "k_envelope": samples_per_step - routed + flagged,
           "k_spectral_envelope": routed, "k_utterance_spectrum": routed, "k_tail_state": routed}
    rep = {}
    for k, (n, ms) in prof.items():
        per_step_s = ms / steps / 1e3
        r = {"launches": n, "ms_per_step": round(ms / steps, 4)}
        if k in own:
            smp = own[k]
            sc = smp * C           # sample-channels this kernel processed per step
            r["samples_per_step"] = smp
            need = {"k_erb_filterbank": 2 * smp + (8 * sc if mode == "filterbank" else 0), "k_envelope": 8 * sc,
                    "k_spectral_envelope": 8 * sc, "k_utterance_spectrum": 2 * smp, "k_tail_state": 0}[k]
            moved = {"k_erb_filterbank": 2 * smp + (8 if mode == "filterbank" else handoff_bytes) * sc,
                     "k_envelope": (handoff_bytes + 8) * sc, "k_spectral_envelope": 8 * sc, "k_utterance_spectrum": 2 * smp,
                     "k_tail_state": 0}[k]
            if smp == 0:
                r["note"] = "skip-only launches: every utterance of the step was served by the spectral kernel (flags walked, no rows)"
            else:
                r["required_GBps"] = round(need / per_step_s / 1e9, 1)
                r["moved_GBps"] = round(moved / per_step_s / 1e9, 1)
                r["required_bytes_per_step"] = need
                if k == "k_erb_filterbank":
                    tf = K1_FLOP_PER_SAMPLE_CHANNEL * sc / per_step_s / 1e12
                    r["f64_TFLOPps"] = round(tf, 2)
                    r["f64_frac_of_peak"] = round(tf / F64_PEAK_TFLOPS, 4)
        rep[k] = r
    return rep


def dsp_run(ctx, ranks, coefs, C, N, workload, fft, steps, warmup, batch, corpus):
    """One DSP workload on every rank; rank 0 gets the merged result (host-side merge, no collective on the data path)."""
    rank, world = ranks.rank, ranks.world
    seed = SEEDS[workload]
    mode = "filterbank" if workload == "cfg2" else "both"
    lpf = 0 if workload == "cfg2" else 50
    if workload in ("cfg5", "cfg5r"):
        all_lens = ragged_lengths(seed, corpus) if workload == "cfg5r" else np.full(corpus, N, np.int64)
        idx = np.arange(corpus)[rank::world]            # files[r::G]
        lens = all_lens[idx]
        waves = synth_corpus(seed, idx, lens, max(1, host_cores() // world))
        # launches of up to 2500 utterances: a rank's shard of the 10 000 splits into equal launches that still fill the
        # chip at 8 ranks (1250 utterances = one launch of 2500 wavefronts); ragged launches of that size give the
        # filterbank's unit queue two units per wave at two waves per SIMD
        batch = batch or 2500
        scaling, utts = "strong", corpus
        label = (f"{workload}: {corpus}-utterance corpus (seed 2029+u, "
                 f"{'U[16000,64000] samples' if workload == 'cfg5r' else f'{N / FS:g} s'} each), rank r owns utterances "
                 f"r::{world}, batches of <= {batch} utterances through the filterbank + envelope + {lpf} Hz LPF call, "
                 f"{C} channels, ENV1 (float64) out; one step = one pass over the corpus")
    else:
        B = batch or {"cfg2": 256, "cfg3": 1000}[workload]
        lens = np.full(B, N, np.int64)
        waves = synth_batch(seed, rank * B, B, N).reshape(-1)
        batch, scaling, utts = B, "weak", B * world
        label = (f"cfg2: {C}-channel gammatone filterbank, batch of {B} x {N / FS:g} s utterances per GPU, GFB (float64) out"
                 if workload == "cfg2" else
                 f"cfg3: filterbank + Hilbert envelope + {lpf} Hz LPF in one call ({fft} FFT), batch of {B} x {N / FS:g} s "
                 f"utterances per GPU, {C} channels, ENV1 (float64) out")
    if ctx is None:     # --dry-run: everything but the device work
        ranks.barrier()
        res = ranks.gather({"elapsed": 0.001 * (rank + 1), "samples_per_step": int(lens.sum()),
                            "utterances": len(lens), "checksum": int(waves.astype(np.int64).sum())})
        if rank != 0:
            return None
        elapsed = max(r["elapsed"] for r in res)
        total = sum(r["samples_per_step"] for r in res)
        return {"value": round(total / FS * steps / elapsed, 1), "scaling": scaling, "workload": label,
                "utterances_per_step": utts, "checksum": sum(r["checksum"] for r in res), "per_rank": res}
    from f2cnn_amd import _lib
    precision = _lib.FFT_F32 if fft == "f32" else _lib.FFT_F64
    job = DspJob(ctx, coefs, C, waves, lens, batch, mode, lpf, precision)
    del waves
    elapsed, prof = timed(ctx, ranks, job.step, steps, warmup)
    flagged = int(ctx.get_option("spectral_flagged")) if mode == "both" else 0
    samples = job.total_samples
    routed_s, flagged_s = job.routing()
    # hand-off bytes per sample-channel, as f2_plan_handoff decides: float32 when the FFT is float32, else float64
    handoff = 4 if fft == "f32" else 8
    mine = {"elapsed": elapsed, "samples_per_step": samples,
            "kernels": dsp_kernel_report(prof, steps, C, samples, mode, handoff, routed_s, flagged_s), "flagged": flagged,
            "routed_samples": routed_s, "flagged_samples": flagged_s}
    resident, nlaunch = job.resident, len(job.batches)
    job.free()
    res = ranks.gather(mine)
    if rank != 0:
        return None
    elapsed = max(r["elapsed"] for r in res)
    step_s = elapsed / steps
    total_samples = sum(r["samples_per_step"] for r in res)
    need_step = (2 + 8 * C) * total_samples                     # 2*fs + 8*C*fs bytes per audio-second
    kern = mine["kernels"]
    kname = max(kern, key=lambda k: kern[k]["ms_per_step"])          # dominant kernel (rank 0's device times)
    samples_via_spectral = sum(r.get("routed_samples", 0) for r in res)
    kd = kern[kname]
    traffic = measured_traffic(workload, batch, C, N, fft)
    roof = {"bound": "hbm", "kernel": kname, "achieved": kd.get("required_GBps"), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kd.get("required_GBps", 0.0) / HBM_PEAK_GBS, 4),
            "traffic": traffic.get(kname) if traffic else None,
            "algorithmic_bytes_per_launch": kd.get("required_bytes_per_step", 0) // max(kd["launches"] // steps, 1),
            "avg_launch_ms": round(kd["ms_per_step"] * steps / kd["launches"], 4), "launches_timed": kd["launches"],
            "moved_GBps": kd.get("moved_GBps"), "moved_frac": round(kd.get("moved_GBps", 0.0) / HBM_PEAK_GBS, 4),
            "step": {"algorithmic_bytes": need_step, "achieved": round(need_step / step_s / 1e9 / world, 1),
                     "frac": round(need_step / step_s / 1e9 / world / HBM_PEAK_GBS, 4),
                     "note": "whole step, per GPU: (2 + 8*C) bytes per sample / wall time per step"},
            "note": "achieved = bytes the kernel must move (SURVEY 8d; a K1->K2 hand-off, where the two-kernel route runs, is "
                    "overhead counted only in moved_GBps) / its mean device time per launch (HIP events on the launch stream)"}
    if traffic:
        roof["traffic_all_kernels"] = traffic
        roof["traffic_over_algorithmic"] = round(sum(traffic.values()) / (need_step / world), 3)
    return {"value": round(total_samples / FS * steps / elapsed, 1), "ms_per_step": round(step_s * 1e3, 4), "steps": steps,
            "scaling": scaling, "lpf": lpf, "mode": mode, "roofline": roof, "kernels": kern,
            "config": {"workload": label, "channels": C, "sample_rate": FS, "lpf_hz": lpf, "utterances_per_step": utts,
                       "launch_batch": batch, "launches_per_step_per_gpu": nlaunch, "outputs_resident_in_hbm": resident,
                       "route": ("k_spectral_envelope (one kernel per row class; k_utterance_spectrum + k_tail_state once per "
                                 "utterance) where it appears in `kernels`, else k_erb_filterbank -> k_envelope"),
                       "utterances_sent_back_by_the_accuracy_guard": sum(r.get("flagged", 0) for r in res),
                       "fraction_of_samples_through_the_spectral_kernel": round(samples_via_spectral / max(total_samples, 1), 4),
                       "parallelism": f"utterance-sharded x{world}, no collective"},
            "per_rank": [{"elapsed_s": round(r["elapsed"], 4), "audio_s_per_step": round(r["samples_per_step"] / FS, 1)}
                         for r in res]}


# ------------------------------------------------------------------------------------------------
# side blocks of the default single-GPU run
# ------------------------------------------------------------------------------------------------
def block_cfg4(ctx, coefs, precision, B, N, steps, warmup, ranks, with_cpu):
    from f2cnn_amd import _lib
    from f2cnn_amd.model import F2CNNModel
    model = F2CNNModel.glorot(7)
    hcnn = model.handle(ctx)
    nb = N - 11 * 160
    waves = synth_batch(SEEDS["cfg4"], ranks.rank * B, B, N)
    offsets = np.arange(B + 1, dtype=np.int64) * N
    d_wave = ctx.malloc(waves.nbytes)
    ctx.h2d(d_wave, waves)
    d_scores, d_labels = ctx.malloc(8 * nb * B), ctx.malloc(nb * B)

    def step():
        ctx.eval_batch(hcnn, d_wave, _lib.WAVE_I16, offsets, coefs, B, 128, False, 0.0, precision, 5, 160, d_scores,
                       d_labels, _lib.MEM_DEVICE)
    elapsed, prof = timed(ctx, ranks, step, steps, warmup)
    for p in (d_wave, d_scores, d_labels):
        ctx.free(p)
    cnn_s = prof["k_cnn_forward"][1] / 1e3
    flop = CNN_FLOP_PER_WINDOW * nb * B * steps
    out = {"workload": f"cfg4: cnn eval end to end (filterbank, envelope, every-sample 11x128 windows, normalise, CNN), "
                       f"{B} x {N / FS:g} s utterances per GPU = {B * nb} windows, Glorot weights seed 7",
           "value": round(B * N / FS * steps / elapsed, 2), "unit": "audio-seconds/s", "steps": steps,
           "ms_per_step": round(elapsed / steps * 1e3, 3),
           "cnn": {"launch_groups": prof["k_cnn_forward"][0], "ms_per_step": round(cnn_s / steps * 1e3, 3),
                   **cnn_accounting(ctx, flop, cnn_s, nb * B * steps)},
           "kernels_ms_per_step": {k: round(ms / steps, 3) for k, (n, ms) in prof.items()}}
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline_cnn(SEEDS["cfg4"], N)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        out["gpu_over_cpu_all_cores_est"] = all_cores_estimate(out["value"], out["cpu_baseline"])
    return out, prof, elapsed, flop


def block_sustained(ctx, coefs, C, N, rank, seconds=6.0):
    """The cfg3 step repeated for `seconds` of wall time without a pause: the rate the chip holds once clocks and
    temperature have settled (the K timed steps of the headline are 0.1 s of GPU time), and a stretch of GPU work long
    enough for an external utilisation sampler to see."""
    from f2cnn_amd import _lib
    B = 1000
    waves = synth_batch(SEEDS["cfg3"], rank * B, B, N).reshape(-1)
    job = DspJob(ctx, coefs, C, waves, np.full(B, N, np.int64), B, "both", 50, _lib.FFT_F32)
    del waves
    for _ in range(3):
        job.step()
    ctx.synchronize()
    steps, t0 = 0, time.perf_counter()
    first = last = None
    while True:
        t1 = time.perf_counter()
        for _ in range(50):
            job.step()
        ctx.synchronize()
        t2 = time.perf_counter()
        steps += 50
        last = (t2 - t1) / 50
        first = first if first is not None else last
        if t2 - t0 >= seconds:
            break
    elapsed = time.perf_counter() - t0
    job.free()
    return {"workload": f"cfg3 (1000 x {N / FS:g} s, {C} channels, LPF 50) repeated for {seconds:g} s of wall time, a device sync "
                        "every 50 steps", "value": round(B * N / FS * steps / elapsed, 1), "unit": "audio-seconds/s", "steps": steps,
            "seconds": round(elapsed, 2), "ms_per_step": round(elapsed / steps * 1e3, 4),
            "ms_per_step_first_50": round(first * 1e3, 4), "ms_per_step_last_50": round(last * 1e3, 4)}


def block_cfg1(ctx, ranks, steps=200, warmup=20):
    """BASELINE configs[0], the reference's own CPU-runnable case: ONE 1 s utterance (seed 1234), 64 channels, filterbank +
    envelope + 50 Hz low-pass, input and output resident in HBM. One step = one fused call followed by a device sync,
    i.e. the latency a caller of a single file sees (the filterbank runs its time-split path here)."""
    from f2cnn_amd import _lib
    from f2cnn_amd.gammatone import filters
    C1, N1 = 64, FS
    coefs = filters.make_erb_filters(FS, filters.centre_freqs(FS, C1, 100))
    wave = synth_utterance(1234, N1)
    offsets = np.array([0, N1], dtype=np.int64)
    d_wave, d_env = ctx.malloc(wave.nbytes), ctx.malloc(8 * C1 * N1)
    ctx.h2d(d_wave, wave)

    def step():
        ctx.filterbank_envelope_fused(d_wave, _lib.WAVE_I16, offsets, coefs, 1, C1, True, 50.0, _lib.FFT_F32, d_env, None,
                                      _lib.MEM_DEVICE)
        ctx.synchronize()
    elapsed, prof = timed(ctx, ranks, step, steps, warmup)
    ctx.free(d_wave)
    ctx.free(d_env)
    return {"workload": "cfg1: one 1 s utterance (seed 1234), 64 channels, fused filterbank + envelope + 50 Hz LPF, one "
                        "call + device sync per step (single-file latency)",
            "value": round(steps * N1 / FS / elapsed, 1), "unit": "audio-seconds/s", "steps": steps,
            "latency_us": round(elapsed / steps * 1e6, 1),
            "kernels_us_per_step": {k: round(ms / steps * 1e3, 1) for k, (n, ms) in prof.items()}}


def _e2e_cpu_one(args):
    """CPU equivalent of `prepare filter` + `prepare envelope` for one file: read, filterbank, save GFB, load GFB,
    envelope, save ENV1 (reference GammatoneFiltering.py:69-83, EnvelopeExtraction.py:101-117) with the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import f2cnn_oracle as orc
    from f2cnn_amd import wavio
    path, lpf = args
    coefs = orc.make_erb_filters(FS, orc.centre_freqs(FS, 128, 100))
    t = time.perf_counter()
    _, wave = wavio.read_audio(path)
    base = os.path.splitext(path)[0]
    np.save(base + ".cpu.GFB.npy", orc.erb_filterbank(wave, coefs))
    np.save(base + ".cpu.ENV1.npy", orc.extract_envelope_from_matrix(np.load(base + ".cpu.GFB.npy"), True, lpf))
    return time.perf_counter() - t


def pcie_link(local=0):
    """Nominal one-direction bandwidth of the PCIe link of HIP device `local`, from sysfs (no GPU call):
    {"gpu", "speed", "width", "GBps"} or None. GT/s x lanes / 8, less the 128b/130b line code (gen 3 and later)."""
    from f2cnn_amd.runtime import gpu_numa_nodes
    gpus = gpu_numa_nodes()
    if not (0 <= local < len(gpus)):
        return None
    bdf = gpus[local][0]
    try:
        speed = open(f"/sys/bus/pci/devices/{bdf}/current_link_speed").read().strip()
        width = int(open(f"/sys/bus/pci/devices/{bdf}/current_link_width").read().strip())
        gts = float(speed.split()[0])
    except (OSError, ValueError, IndexError):
        return None
    return {"gpu": bdf, "speed": speed, "width": width, "GBps": round(gts * width / 8 * (128 / 130 if gts >= 8 else 0.8), 1)}


def block_end_to_end(n_files, cpu_files, with_cpu, local=0):
    """File level, host I/O and PCIe included: n_files x 1 s SPHERE files on tmpfs through the CLI drivers
    (`prepare filter` + `prepare envelope`, and the one-pass `prepare features`), and the oracle + numpy.save
    equivalent on the host cores beside it. Never the bench `value`."""
    import contextlib
    import io
    import multiprocessing as mp
    import shutil
    import tempfile
    from f2cnn_amd import cli, config, wavio
    base = tempfile.mkdtemp(prefix="f2bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    cwd = os.getcwd()
    try:
        os.chdir(base)
        config.write_default()
        os.makedirs("resources/f2cnn/TEST")
        waves = synth_batch(SEEDS["cfg5"], 0, n_files, FS)
        names = [f"resources/f2cnn/TEST/DR1.S{i:04d}.SA1.WAV" for i in range(n_files)]
        for name, w in zip(names, waves):
            wavio.write_sphere(name, FS, w)

        def run(argv):
            t = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                if cli.main(argv) != 0:
                    raise RuntimeError(f"{argv} failed")
            return time.perf_counter() - t
        run(["prepare", "features", "--cutoff", "50"])          # warm-up (twiddles, allocations, page cache)
        t1, t2 = run(["prepare", "filter"]), run(["prepare", "envelope", "--cutoff", "50"])
        t3 = run(["prepare", "features", "--cutoff", "50"])
        out = {"files": n_files, "audio_s": n_files, "where": base.split("/")[1],
               "prepare_filter_s": round(t1, 3), "prepare_envelope_s": round(t2, 3),
               "two_commands_audio_s_per_s": round(n_files / (t1 + t2), 1),
               "prepare_features_s": round(t3, 3), "one_pass_audio_s_per_s": round(n_files / t3, 1),
               "npy_bytes_per_pass": n_files * 2 * 128 * FS * 8}
        # this block's roofline is the PCIe link: every .npy byte crosses it once (device -> host), the waves the other way
        out["npy_GBps"] = round(out["npy_bytes_per_pass"] / t3 / 1e9, 1)
        out["npy_GBps_two_commands"] = round((out["npy_bytes_per_pass"] + n_files * 128 * FS * 8) / (t1 + t2) / 1e9, 1)
        link = pcie_link(local)
        out["pcie_link"] = link
        if link:
            out["npy_frac_of_pcie"] = round(out["npy_GBps"] / link["GBps"], 3)
            out["note"] = ("bound by the PCIe link and numpy.save, not by the kernels: npy_GBps = .GFB.npy + .ENV1.npy bytes of one "
                           "`prepare features` pass / its wall time; two commands also read the .GFB.npy files back")
        if with_cpu:
            cores = host_cores()
            cpu_files = min(cpu_files, n_files)
            with mp.get_context("spawn").Pool(cores) as pool:
                pool.map(_cpu_one, [(1, i, 256, 8, 50, "both") for i in range(cores)])
                t0 = time.perf_counter()
                pool.map(_e2e_cpu_one, [(os.path.join(base, n), 50) for n in names[:cpu_files]], chunksize=1)
                wall = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(cpu_files / wall, 2), "unit": "audio-seconds/s", "cores": cores,
                                   "kind": "port", "sample": f"{cpu_files} of the same files: read, oracle filterbank, "
                                   f"numpy.save GFB, numpy.load, oracle envelope (LPF 50), numpy.save ENV1, one file per "
                                   f"task in a {cores}-process pool; {wall:.1f} s wall"}
            out["cpu_baseline"]["host"] = host_core_facts(cores)
            out["two_commands_over_cpu"] = round(out["two_commands_audio_s_per_s"] / out["cpu_baseline"]["value"], 1)
            out["two_commands_over_cpu_all_cores_est"] = all_cores_estimate(out["two_commands_audio_s_per_s"], out["cpu_baseline"])
        return out
    finally:
        os.chdir(cwd)
        shutil.rmtree(base, ignore_errors=True)


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5", "cfg5r"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per launch (default: the config's)")
    ap.add_argument("--corpus", type=int, default=CORPUS, help="utterances in the cfg5 corpus")
    ap.add_argument("--channels", type=int, default=128)
    ap.add_argument("--samples", type=int, default=FS, help="samples per utterance")
    ap.add_argument("--fft", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64)
    ap.add_argument("--no-blocks", action="store_true", help="skip the side blocks of the default single-GPU run")
    ap.add_argument("--e2e-files", type=int, default=256)
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: ranks only shard + generate the corpus and meet for the merge (CPU test of the N > 1 path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly one JSON line: everything else a rank prints (gloo's connection notes, the file drivers'
    # progress lines) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if os.environ.get("F2CNN_BENCH_ONE_DEVICE") == "1":      # rehearsal of the multi-rank path on one GPU
        local = 0
    args.gpus = world
    workload = args.workload or ("cfg3" if world == 1 else "cfg5")

    if args.dry_run:
        ranks = Ranks(rank, world)
        run = dsp_run(None, ranks, None, args.channels, args.samples, workload, args.fft, args.steps, args.warmup,
                      args.batch, args.corpus)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, **run}), file=json_out, flush=True)
        ranks.close()
        return

    # (before the first GPU call of this process) stay on the cores of the NUMA node this rank's GPU hangs off
    from f2cnn_amd.runtime import pin_to_gpu_numa_node
    placement = pin_to_gpu_numa_node(local)
    from f2cnn_amd import _lib
    from f2cnn_amd.gammatone import filters
    ranks = Ranks(rank, world)
    ctx = _lib.Context(local)
    C, N = args.channels, args.samples
    coefs = filters.make_erb_filters(FS, filters.centre_freqs(FS, C, 100))
    precision = _lib.FFT_F32 if args.fft == "f32" else _lib.FFT_F64
    seed = SEEDS[workload]
    with_cpu = not args.no_cpu_baseline
    out = None

    if workload == "cfg4":
        if C != 128:
            raise SystemExit("cfg4 uses the 11 x 128 network")
        if args.steps == 20 and args.warmup == 3:
            args.steps, args.warmup = 3, 1
        B = args.batch or 8
        blk, prof, elapsed, flop = block_cfg4(ctx, coefs, precision, B, N, args.steps, args.warmup, ranks,
                                              with_cpu and rank == 0)
        res = ranks.gather({"elapsed": elapsed, "audio_s": B * N / FS * args.steps})
        if rank == 0:
            elapsed = max(r["elapsed"] for r in res)
            cnn_s = prof["k_cnn_forward"][1] / 1e3
            out = {"metric": "audio-seconds/sec through filterbank+envelope+CNN (HIP, 1 MI355X per rank)",
                   "value": round(sum(r["audio_s"] for r in res) / elapsed, 2), "unit": "audio-seconds/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": f"f64 IIR + {args.fft} FFT + f32 CNN (conv2-conv4, dense1 products as 3 split-fp16 MFMAs, f32 accumulate)", "data": "synthetic",
                   "config": {"workload": blk["workload"], "batch_per_gpu": B, "channels": C, "samples_per_utterance": N,
                              "sample_rate": FS, "lpf_hz": 0, "parallelism": f"utterance-sharded x{world}, no collective"},
                   "roofline": {"bound": "mfma", "kernel": "k_cnn_forward",
                                "achieved": round(blk["cnn"].get("issued_TFLOPps", blk["cnn"]["TFLOPps"]) * 1e3, 1),
                                "peak": blk["cnn"]["peak_TFLOPps"] * 1e3, "unit": "GFLOP/s",
                                "frac": blk["cnn"]["frac"], "traffic": None, "arithmetic": blk["cnn"]["arithmetic"],
                                "algorithmic_GFLOPps": round(flop / cnn_s / 1e9, 1),
                                "algorithmic_flop_per_launch": flop // prof["k_cnn_forward"][0],
                                "avg_launch_ms": round(cnn_s * 1e3 / prof["k_cnn_forward"][0], 4),
                                "launches_timed": prof["k_cnn_forward"][0]},
                   "kernels": blk["kernels_ms_per_step"], "cpu_baseline": blk.get("cpu_baseline")}
            if out["cpu_baseline"]:
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                out["gpu_over_cpu_all_cores_est"] = all_cores_estimate(out["value"], out["cpu_baseline"])
    else:
        if args.steps == 20 and args.warmup == 3 and workload == "cfg5r":
            args.steps, args.warmup = 3, 1
        run = dsp_run(ctx, ranks, coefs, C, N, workload, args.fft, args.steps, args.warmup, args.batch, args.corpus)
        if rank == 0:
            out = {"metric": "audio-seconds/sec through filterbank" + ("" if workload == "cfg2" else "+envelope")
                             + " (HIP, 1 MI355X per rank); the +CNN figure of BASELINE's metric is value_with_cnn / blocks.cfg4",
                   "value": run["value"], "unit": "audio-seconds/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": run["ms_per_step"], "higher_is_better": True,
                   "scaling": run["scaling"], "vs_baseline": None,
                   "dtype": "f64" if workload == "cfg2" else f"f64 filter state + f64 utterance FFT + {args.fft} row FFT", "data": "synthetic",
                   "config": run["config"], "roofline": run["roofline"], "kernels": run["kernels"],
                   "per_rank": run["per_rank"]}
            if with_cpu:
                n_cpu = N if workload != "cfg5r" else 40000
                out["cpu_baseline"] = cpu_baseline(seed, n_cpu, C, run["lpf"], run["mode"], args.cpu_sample)
                out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                out["gpu_over_cpu_all_cores_est"] = all_cores_estimate(out["value"], out["cpu_baseline"])

    # side blocks: the other configurations in the same line (default runs only)
    blocks = {}
    if args.workload is None and not args.no_blocks:
        def brief(r):
            return {"workload": r["config"]["workload"], "value": r["value"], "unit": "audio-seconds/s",
                    "steps": r["steps"], "ms_per_step": r["ms_per_step"], "scaling": r["scaling"],
                    "step_frac_of_hbm_peak": r["roofline"]["step"]["frac"], "kernels": r["kernels"]}
        if world == 1:
            blocks["cfg3_sustained"] = block_sustained(ctx, coefs, C, N, rank)
            blocks["cfg3_fft_f64"] = brief(dsp_run(ctx, ranks, coefs, C, N, "cfg3", "f64", 5, 1, args.batch, args.corpus))
            # BASELINE configs[1]: the filterbank alone, float64 GFB out (`prepare filter` with the matrix left on the device)
            blocks["cfg2"] = brief(dsp_run(ctx, ranks, coefs, C, N, "cfg2", "f32", 20, 5, None, args.corpus))
            blocks["cfg5"] = brief(dsp_run(ctx, ranks, coefs, C, N, "cfg5", "f32", 2, 1, None, args.corpus))
        r = dsp_run(ctx, ranks, coefs, C, N, "cfg5r", "f32", 2, 1, None, args.corpus)
        if rank == 0:
            blocks["cfg5_ragged"] = brief(r)
        if world == 1:
            blocks["cfg1"] = block_cfg1(ctx, ranks)
        if world == 1 and C == 128:
            blocks["cfg4"] = block_cfg4(ctx, coefs, _lib.FFT_F32, 8, N, 2, 1, ranks, with_cpu)[0]
            # the same with every product of the CNN in exact float32 (v_mfma_f32_32x32x2_f32): the figure without the
            # split asterisk (north star: identical labels)
            ctx.set_option("cnn_f16x3", 0)
            try:
                b32 = block_cfg4(ctx, coefs, _lib.FFT_F32, 8, N, 1, 1, ranks, False)[0]
            finally:
                ctx.set_option("cnn_f16x3", 1)
            blocks["cfg4_cnn_f32"] = {k: b32[k] for k in ("workload", "value", "unit", "steps", "ms_per_step", "cnn")}
        if world == 1:
            ctx.synchronize()
            try:
                blocks["end_to_end"] = block_end_to_end(args.e2e_files, 128, with_cpu, local)
            except Exception as exc:       # a full tmpfs must not lose the kernel numbers
                blocks["end_to_end"] = {"error": repr(exc)}
        if rank == 0 and out is not None:
            out["blocks"] = blocks
            if "cfg4" in blocks and "value" in blocks["cfg4"]:
                out["value_with_cnn"] = blocks["cfg4"]["value"]      # BASELINE's metric string: filterbank+envelope+CNN
            # the side blocks' headline figures as top-level scalars (a reader that keeps only those still sees them)
            if "cfg4_cnn_f32" in blocks:
                out["value_with_cnn_f32"] = blocks["cfg4_cnn_f32"]["value"]
            if "cfg3_fft_f64" in blocks:
                out["value_fft_f64"] = blocks["cfg3_fft_f64"]["value"]
            if "cfg2" in blocks:
                out["value_filterbank_only"] = blocks["cfg2"]["value"]
            if "cfg5_ragged" in blocks:
                out["value_ragged"] = blocks["cfg5_ragged"]["value"]
            if "cfg1" in blocks:
                out["cfg1_latency_us"] = blocks["cfg1"]["latency_us"]
            if "cfg3_sustained" in blocks:
                out["value_sustained"] = blocks["cfg3_sustained"]["value"]
            if "cfg4" in blocks and "cnn" in blocks["cfg4"]:
                out["cnn_issued_frac_of_bf16_peak"] = blocks["cfg4"]["cnn"].get("frac")
                out["cnn_algorithmic_frac_of_bf16_peak"] = blocks["cfg4"]["cnn"].get("algorithmic_frac_of_bf16_peak")
    if rank == 0 and out is not None:
        out["host_placement"] = placement or "not pinned (single NUMA node or no sysfs answer)"
        # one workload for every point of a scaling curve: the cfg5 corpus (what N > 1 reports as `value`) also at N = 1
        if workload == "cfg5":
            out["scale"] = {"workload": "cfg5", "value": out["value"], "unit": "audio-seconds/s", "scaling": "strong"}
        elif "cfg5" in blocks:
            out["scale"] = {"workload": "cfg5", "value": blocks["cfg5"]["value"], "unit": "audio-seconds/s", "scaling": "strong"}
        print(json.dumps(out), file=json_out, flush=True)
    ctx.close()
    ranks.close()


if __name__ == "__main__":
    main()
