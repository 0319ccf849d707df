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


def _pci_numa_node(sysfs, bdf):
    try:
        return int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD GPU of the host in the order the ROCm runtime enumerates them, read from sysfs without touching
    the GPU: [(pci address, node)], node -1 where the platform reports none. The order comes from the KFD topology
    (/sys/class/kfd/kfd/topology/nodes/<i>/properties: nodes with SIMDs are GPUs, `domain` + `location_id` give the PCI
    function) - HIP device i is the i-th of them; hosts without that tree fall back to the AMD display / accelerator
    functions in PCI address order, which is NOT guaranteed to be the enumeration order (round-3 advisor finding):
    pin_to_gpu_numa_node only trusts it when there is a single candidate node anyway."""
    kfd = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    out = []
    try:
        nodes = sorted((int(n) for n in os.listdir(kfd) if n.isdigit()))
    except OSError:
        nodes = []
    for n in nodes:
        props = {}
        try:
            for line in open(os.path.join(kfd, str(n), "properties")):
                k, _, v = line.strip().partition(" ")
                props[k] = v
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue                                   # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (KeyError, ValueError):
            continue
        bdf = "{:04x}:{:02x}:{:02x}.{:x}".format(dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 0x7)
        out.append((bdf, _pci_numa_node(sysfs, bdf)))
    if out:
        return out
    base = os.path.join(sysfs, "bus", "pci", "devices")
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
        out.append((bdf, _pci_numa_node(sysfs, bdf)))
    return out


def _kfd_order_known(sysfs):
    return os.path.isdir(os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes"))


def _render_nodes(dev):
    """number of DRM render nodes this process can see (/dev/dri/renderD*), or None if the directory cannot be read"""
    try:
        return sum(1 for n in os.listdir(os.path.join(dev, "dri")) if n.startswith("renderD"))
    except OSError:
        return None


def pin_to_gpu_numa_node(local=None, sysfs="/sys", apply=True, dev="/dev"):
    """Restrict this process to the cores of the NUMA node its GPU hangs off, BEFORE the first GPU call: the staging
    buffers it then allocates (first touch) and its reader / writer threads stay next to the device's PCIe root. With
    eight ranks on a two-socket host the alternative is that half of them copy across the socket link.
    `local` = index among the visible GPUs (default: local_device()); ROCR_VISIBLE_DEVICES filters the runtime's list first
    and HIP_VISIBLE_DEVICES then indexes what is left - both are applied, in that order, when they list plain indices.
    Returns {"gpu", "numa_node", "cpus"} or None - and changes nothing - whenever the answer is not certain: no sysfs entry,
    a visibility variable that is not a list of indices (UUIDs), an index out of range, affinity calls unavailable, an
    enumeration order that had to be guessed from PCI addresses while the GPUs sit on different nodes (a wrong guess would
    pin the rank to the far socket, the copies this function exists to avoid), CUDA_VISIBLE_DEVICES / GPU_DEVICE_ORDINAL
    set (HIP honours them too, with rules of their own), or a container whose device cgroup hides some of the GPUs the KFD
    topology lists (fewer /dev/dri/renderD* nodes than KFD GPU nodes: the runtime skips the hidden ones, the sysfs tree
    still counts them - round-4 advisor finding)."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    if os.environ.get("CUDA_VISIBLE_DEVICES") or os.environ.get("GPU_DEVICE_ORDINAL"):
        return None
    local = local_device() if local is None else int(local)
    gpus = gpu_numa_nodes(sysfs)
    if not _kfd_order_known(sysfs) and len({node for _, node in gpus}) > 1:
        return None
    if _kfd_order_known(sysfs) and len({node for _, node in gpus}) > 1 and _render_nodes(dev) != len(gpus):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        visible = os.environ.get(var)
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
