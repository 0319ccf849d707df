"""Multi-GPU sharding: utterances (files) are independent, so rank r of G simply takes files[r::G]
(reference analogue: the multiprocessing.Pool over files at
scripts/processing/GammatoneFiltering.py:122-125). No collective is involved."""
import os


def rank_world():
    """(rank, world) from the torchrun-style environment, or F2CNN_RANK/F2CNN_WORLD; (0,1) otherwise."""
    rank = int(os.environ.get("F2CNN_RANK", os.environ.get("RANK", "0")))
    world = int(os.environ.get("F2CNN_WORLD", os.environ.get("WORLD_SIZE", "1")))
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return rank, world


def shard(items, rank, world):
    """Round-robin shard of a sorted list: every item is owned by exactly one rank."""
    return list(items)[rank::world]


def shard_for_rank(items):
    r, w = rank_world()
    return shard(items, r, w)


def local_device():
    """Device index for this process: LOCAL_RANK under torchrun, else $F2CNN_DEVICE, else 0."""
    return int(os.environ.get("F2CNN_DEVICE", os.environ.get("LOCAL_RANK", "0")))


# ---- host-side placement of a rank: the cores next to its GPU ----
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD GPU of the host, in PCI address order (the order HIP enumerates them in), read from sysfs
    without touching the GPU: [(pci address, node)], node -1 where the platform reports none."""
    base = os.path.join(sysfs, "bus", "pci", "devices")
    out = []
    try:
        names = sorted(os.listdir(base))
    except OSError:
        return out
    for bdf in names:
        dev = os.path.join(base, bdf)
        try:
            vendor = open(os.path.join(dev, "vendor")).read().strip().lower()
            cls = open(os.path.join(dev, "class")).read().strip().lower()
        except OSError:
            continue
        # display controllers (0x03xxxx) and processing accelerators (0x12xxxx) of vendor 0x1002
        if vendor != "0x1002" or not (cls.startswith("0x03") or cls.startswith("0x12")):
            continue
        try:
            node = int(open(os.path.join(dev, "numa_node")).read().strip())
        except (OSError, ValueError):
            node = -1
        out.append((bdf, node))
    return out


def pin_to_gpu_numa_node(local=None, sysfs="/sys", apply=True):
    """Restrict this process to the cores of the NUMA node its GPU hangs off, BEFORE the first GPU call: the staging
    buffers it then allocates (first touch) and its reader / writer threads stay next to the device's PCIe root. With
    eight ranks on a two-socket host the alternative is that half of them copy across the socket link.
    `local` = index among the visible GPUs (default: local_device(), mapped through HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES when they list plain indices). Returns {"gpu", "numa_node", "cpus"} or None when the platform
    gives no answer (one node, no sysfs entry, affinity calls unavailable): then nothing is changed."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    local = local_device() if local is None else int(local)
    gpus = gpu_numa_nodes(sysfs)
    visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if visible:
        try:
            idx = [int(v) for v in visible.split(",") if v.strip() != ""]
            gpus = [gpus[i] for i in idx]
        except (ValueError, IndexError):
            return None
    if not (0 <= local < len(gpus)):
        return None
    bdf, node = gpus[local]
    if node < 0:
        return None
    try:
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", "node{}".format(node), "cpulist")).read())
    except (OSError, ValueError):
        return None
    allowed = cpus & set(os.sched_getaffinity(0))
    if not allowed:
        return None
    if apply and allowed != set(os.sched_getaffinity(0)):
        os.sched_setaffinity(0, allowed)
    return {"gpu": bdf, "numa_node": node, "cpus": len(allowed)}
