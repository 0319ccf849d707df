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
