import os
"""CPU-only tests of the host-side mirror (filter design etc.) against the golden vectors."""
import numpy as np
import pytest

from f2cnn_amd.gammatone import filters


@pytest.mark.parametrize("C", [8, 64, 128])
def test_filter_design_matches_reference(golden, C):
    cf = filters.centre_freqs(16000, C, 100)
    np.testing.assert_allclose(cf, golden[f"g1_cf_{C}"], rtol=1e-14, atol=0)
    co = filters.make_erb_filters(16000, cf)
    ref = golden[f"g1_coefs_{C}"]
    assert co.shape == ref.shape
    np.testing.assert_allclose(co, ref, rtol=1e-12, atol=0)
    assert np.all(co[:, 5] == 0) and np.all(co[:, 6] == 1)


def test_erb_space_defaults():
    assert filters.erb_space().shape == (100,)
    assert abs(filters.erb_point(100, 8000, 1) - 100) < 1e-9 and abs(filters.erb_point(100, 8000, 0) - 8000) < 1e-9


def test_rank_placement_from_a_sysfs_tree(tmp_path, monkeypatch):
    """pin_to_gpu_numa_node reads the GPU -> NUMA node -> cpulist chain from sysfs (here: a fabricated two-socket tree
    with four GPUs, a network card and a GPU without a node) and never touches the device. The enumeration order is the
    KFD topology's, not the PCI address order (round-3 advisor finding); without that tree a multi-node host is left alone."""
    import os
    from f2cnn_amd import runtime

    def dev(bdf, vendor, cls, node):
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "class").write_text(cls + "\n")
        if node is not None:
            (d / "numa_node").write_text(str(node) + "\n")
    dev("0000:05:00.0", "0x1002", "0x120000", 0)
    dev("0000:15:00.0", "0x1002", "0x120000", 0)
    dev("0000:85:00.0", "0x1002", "0x120000", 1)
    dev("0000:95:00.0", "0x1002", "0x038000", None)
    dev("0000:01:00.0", "0x15b3", "0x020000", 0)          # not a GPU
    have = sorted(os.sched_getaffinity(0))
    half = max(1, len(have) // 2)
    for node, cpus in ((0, have[:half]), (1, have[half:] or have[:1])):
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("GPU_DEVICE_ORDINAL", raising=False)
    devfs = tmp_path / "devfs"
    (devfs / "dri").mkdir(parents=True)
    for k in range(4):
        (devfs / "dri" / f"renderD{128 + k}").write_text("")
    import functools
    pin = functools.partial(runtime.pin_to_gpu_numa_node, sysfs=str(tmp_path), apply=False, dev=str(devfs))
    # no KFD topology: PCI address order is only a guess, and the GPUs sit on two nodes -> nothing is pinned
    assert [n for _, n in runtime.gpu_numa_nodes(str(tmp_path))] == [0, 0, 1, -1]
    assert pin(2) is None

    # the runtime's own order: two CPU nodes, then the GPUs - deliberately NOT in PCI address order
    def kfd(index, simds, bdf=None):
        d = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes" / str(index)
        d.mkdir(parents=True)
        text = "cpu_cores_count {}\nsimd_count {}\n".format(0 if simds else 32, simds)
        if bdf:
            dom, bus, rest = bdf.split(":")
            devn, fn = rest.split(".")
            text += "domain {}\nlocation_id {}\n".format(int(dom, 16), (int(bus, 16) << 8) | (int(devn, 16) << 3) | int(fn, 16))
        (d / "properties").write_text(text)
    kfd(0, 0)
    kfd(1, 0)
    for index, bdf in ((2, "0000:85:00.0"), (3, "0000:05:00.0"), (4, "0000:95:00.0"), (5, "0000:15:00.0")):
        kfd(index, 1024, bdf)
    assert runtime.gpu_numa_nodes(str(tmp_path)) == [("0000:85:00.0", 1), ("0000:05:00.0", 0), ("0000:95:00.0", -1), ("0000:15:00.0", 0)]
    got = pin(0)
    assert got == {"gpu": "0000:85:00.0", "numa_node": 1, "cpus": len(have[half:] or have[:1])}
    assert pin(2) is None        # no node reported
    assert pin(7) is None        # no such GPU
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3,0")
    assert pin(0)["numa_node"] == 0
    assert pin(1)["numa_node"] == 1
    # both variables set: ROCR_VISIBLE_DEVICES filters the runtime's list first, HIP_VISIBLE_DEVICES indexes the rest
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,3,0")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2")
    assert pin(0)["gpu"] == "0000:85:00.0"
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-deadbeef")                                # not an index list: no answer
    assert pin(0) is None
    # HIP also honours CUDA_VISIBLE_DEVICES / GPU_DEVICE_ORDINAL, and a container's device cgroup can hide GPUs the KFD
    # tree still lists: either way the index -> GPU mapping is not certain, so nothing is pinned
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert pin(0)["gpu"] == "0000:85:00.0"
    for var in ("CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        monkeypatch.setenv(var, "1")
        assert pin(0) is None
        monkeypatch.delenv(var)
    os.remove(devfs / "dri" / "renderD131")
    assert pin(0) is None
    assert runtime.pin_to_gpu_numa_node(0, sysfs=str(tmp_path), apply=False, dev=str(tmp_path / "nowhere")) is None
    assert runtime._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert sorted(os.sched_getaffinity(0)) == have                                          # apply=False changed nothing


def test_shared_output_markers_carry_the_launch_token(tmp_path, monkeypatch):
    """`prepare input` with several ranks: the marker files are named after the launch (a dead run's
    `ready` marker must not send a rank into the old output file), ranks that disagree about the token fail at once with a
    message naming F2CNN_RUN_ID instead of waiting out the timeout, and the parent's pid is part of the token only where all
    ranks provably share it."""
    import time
    import numpy as np
    import pytest
    from f2cnn_amd.scripts.processing import InputGenerator as ig
    target = str(tmp_path / "input_data.npy")
    monkeypatch.setenv("F2CNN_RUN_ID", "old/run")
    old = ig._shared_output(target, (2, 11, 4), 0, 2)
    old[:] = 7
    del old
    old_marker = ig._marker(target, "ready", 0)
    assert os.path.exists(old_marker) and "old_run" in old_marker
    monkeypatch.setenv("F2CNN_RUN_ID", "new-run")
    # a marker under another token that appeared as this rank started: the launch's ranks disagree - loud and immediate
    t0 = time.time()
    with pytest.raises(RuntimeError, match="F2CNN_RUN_ID"):
        ig._shared_output(target, (3, 11, 4), 1, 2, timeout=30.0)
    assert time.time() - t0 < 5.0
    # the same marker left by a launch that died a while ago: not this launch's, rank 1 waits for rank 0 of ITS launch
    os.utime(old_marker, (time.time() - 60, time.time() - 60))
    with pytest.raises(TimeoutError, match="F2CNN_RUN_ID"):
        ig._shared_output(target, (3, 11, 4), 1, 2, timeout=0.3)
    new = ig._shared_output(target, (3, 11, 4), 0, 2)
    assert new.shape == (3, 11, 4)
    mine = ig._shared_output(target, (3, 11, 4), 1, 2, timeout=5.0)
    mine[1] = 1.0
    mine.flush()
    assert np.load(target).shape == (3, 11, 4) and np.load(target)[1].min() == 1.0
    with pytest.raises(TimeoutError):          # a marker with another shape is not this launch's announcement either
        ig._shared_output(target, (4, 11, 4), 1, 2, timeout=0.3)
    # torchrun without --rdzv-id exports the same run id ('none') and port for every launch: on ONE node (all ranks children
    # of one launcher) the launcher's pid separates the launches ...
    os.remove(old_marker)
    os.remove(ig._marker(target, "ready", 0))
    monkeypatch.delenv("F2CNN_RUN_ID")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setattr(os, "getppid", lambda: 4242)
    crashed = ig._run_token()
    dead = ig._shared_output(target, (3, 11, 4), 0, 2)      # a launch that dies with its `ready` marker in place
    del dead
    os.utime(ig._marker(target, "ready", 0), (time.time() - 60, time.time() - 60))
    monkeypatch.setattr(os, "getppid", lambda: 4343)
    assert ig._run_token() != crashed and crashed.endswith("_p4242")
    with pytest.raises(TimeoutError):          # same shape, same run id, same port - but not this launch's marker
        ig._shared_output(target, (3, 11, 4), 1, 2, timeout=0.3)
    # ... while ranks on several nodes (or behind per-rank wrapper shells) have different parents: launcher id alone
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert ig._run_token() == "none"
    monkeypatch.setattr(os, "getppid", lambda: 5555)
    assert ig._run_token() == "none"
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert ig._run_token() == "none"


def test_shared_output_rejects_a_leftover_marker_of_the_same_token(tmp_path, monkeypatch):
    """Same token, dead launch (several nodes, no F2CNN_RUN_ID: nothing in the token tells the launches apart): the `ready`
    marker says WHICH file it announced. A marker that is too old is ignored; one whose file has been replaced since is
    ignored; and a rank that mapped the leftover's file before rank 0 replaced it finds out when it finishes."""
    import time
    import numpy as np
    import pytest
    from f2cnn_amd.scripts.processing import InputGenerator as ig
    target = str(tmp_path / "input_data.npy")
    monkeypatch.setenv("F2CNN_RUN_ID", "same")
    dead = ig._shared_output(target, (3, 11, 4), 0, 2)
    del dead
    marker = ig._marker(target, "ready", 0)
    leftover = open(marker).read()
    # (a) older than the window
    os.utime(marker, (time.time() - 600, time.time() - 600))
    with pytest.raises(TimeoutError):
        ig._shared_output(target, (3, 11, 4), 1, 2, timeout=0.3)
    # (b) fresh, but the file it describes is gone (rank 0 of the new launch has replaced it, its marker is not there yet)
    os.utime(marker, None)
    fresh = np.lib.format.open_memmap(target + ".new", mode="w+", dtype=np.float32, shape=(3, 11, 4))
    del fresh
    os.replace(target + ".new", target)
    with pytest.raises(TimeoutError):
        ig._shared_output(target, (3, 11, 4), 1, 2, timeout=0.3)
    # (c) rank 1 maps the file of a fresh leftover, then rank 0 of the new launch starts over: rank 1 must not report success
    first = ig._shared_output(target, (3, 11, 4), 0, 2)
    del first
    mine = ig._shared_output(target, (3, 11, 4), 1, 2, timeout=5.0)
    mine[:] = 1.0
    mine.flush()
    del mine
    again = ig._shared_output(target, (3, 11, 4), 0, 2)
    del again
    assert open(marker).read() != leftover
    with pytest.raises(RuntimeError, match="F2CNN_RUN_ID"):
        ig._finish_shared_output(target, str(tmp_path / "backup.npy"), 1, 2)
    assert not os.path.exists(ig._marker(target, "done", 1))
