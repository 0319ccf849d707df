"""N > 1 path on CPU: two gloo ranks shard a corpus the way bench.py / the file drivers do (utterances are
independent, no data-path collective); the ranks only agree on timing through a MAX all-reduce."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import f2cnn_oracle as orc
    from f2cnn_amd import runtime
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert runtime.rank_world() == (rank, world)
    files = sorted(f"resources/f2cnn/TEST/DR1.S{i:02d}.SA1.WAV" for i in range(9))
    mine = runtime.shard_for_rank(files)
    # each rank processes its own utterances (here with the oracle, there is no GPU): checksums of ENV rows
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 8, 100))
    sums = {}
    for f in mine:
        idx = files.index(f)
        wave = bench.synth_batch(2029, idx, 1, 800)[0]
        sums[f] = float(orc.filter_and_envelope(wave, coefs, True, 50).sum())
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, mine, sums, float(t.item())))
    dist.destroy_process_group()


def test_two_ranks_cover_the_corpus_once():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    all_files = sorted(res[0][1] + res[1][1])
    assert len(all_files) == 9 and len(set(all_files)) == 9
    assert all(r[3] == 1.5 for r in res)                      # MAX over ranks, as bench.py times a step
    # rank-local synthetic utterances do not depend on the sharding: same checksum as a single-rank run
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import f2cnn_oracle as orc
    import bench
    coefs = orc.make_erb_filters(16000, orc.centre_freqs(16000, 8, 100))
    merged = {**res[0][2], **res[1][2]}
    files = sorted(merged)
    for i, f in enumerate(files):
        wave = bench.synth_batch(2029, i, 1, 800)[0]
        assert np.isclose(merged[f], orc.filter_and_envelope(wave, coefs, True, 50).sum(), rtol=1e-12)


def _run_bench(*args, launcher=None):
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(launcher),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), *args]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines           # stdout carries the JSON line and nothing else
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_merges_on_the_host():
    """`python bench.py --gpus 2` as the driver types it: the parent starts two fresh rank processes, they shard the
    cfg5 corpus r::G, meet over gloo and rank 0 prints one merged line (--dry-run: everything but the device work)."""
    import bench
    one = _run_bench("--dry-run", "--gpus", "1", "--workload", "cfg5", "--corpus", "11", "--samples", "640", "--steps", "2")
    two = _run_bench("--dry-run", "--gpus", "2", "--corpus", "11", "--samples", "640", "--steps", "2")   # default N>1 = cfg5
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["workload"].startswith("cfg5:")
    assert [r["utterances"] for r in two["per_rank"]] == [6, 5]            # files[r::G]
    # strong scaling: the same corpus whatever the number of ranks
    assert two["checksum"] == one["checksum"] == int(bench.synth_batch(2029, 0, 11, 640).astype(np.int64).sum())
    # value = all ranks' audio-seconds / MAX elapsed over ranks
    total_s = sum(r["samples_per_step"] for r in two["per_rank"]) / 16000 * 2
    assert abs(two["value"] - total_s / max(r["elapsed"] for r in two["per_rank"])) < 0.06


def test_bench_under_torchrun_takes_the_same_path():
    two = _run_bench("--dry-run", "--gpus", "2", "--workload", "cfg5r", "--corpus", "9", "--steps", "1", launcher=2)
    import bench
    lens = bench.ragged_lengths(2029, 9)
    assert two["n_gpus"] == 2 and [r["samples_per_step"] for r in two["per_rank"]] == [int(lens[0::2].sum()), int(lens[1::2].sum())]
    assert lens.min() >= 16000 and lens.max() <= 64000
