"""File I/O pipeline around the GPU kernels.

The reference hides disk latency behind one process per core (GammatoneFiltering.py:122-125,
EnvelopeExtraction.py:145-149); here one process drives one GPU, so the reads of the next batch and the `.npy`
writes of the previous one run on threads (NumPy file I/O releases the GIL) while the GPU works on the current
batch. Results are written by the caller's `save` exactly as the reference names and formats them.
"""
import os
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy


def run_batches(items, load, compute, save, batch=32, readers=4, writers=None, max_pending_writes=64):
    """items -> load(item) [threads] -> compute(list of (item, loaded)) -> iterable of (item, result)
    -> save(item, result) [threads]. Order of `compute` calls follows `items`; returns the number of items saved."""
    items = list(items)
    if not items:
        return 0
    if writers is None:      # the .npy writes (16.4 MB per audio-second and stage) are the slowest stage
        writers = max(4, min(8, (os.cpu_count() or 8) // 2))
    done = 0
    with ThreadPoolExecutor(readers) as rpool, ThreadPoolExecutor(writers) as wpool:
        def submit_loads(chunk):
            return [(it, rpool.submit(load, it)) for it in chunk]
        chunks = [items[s:s + batch] for s in range(0, len(items), batch)]
        pending = submit_loads(chunks[0])
        writes = []
        for ci in range(len(chunks)):
            loaded = [(it, f.result()) for it, f in pending]
            pending = submit_loads(chunks[ci + 1]) if ci + 1 < len(chunks) else []
            for it, res in compute(loaded):
                writes.append(wpool.submit(save, it, res))
                done += 1
            while len(writes) > max_pending_writes:     # bound the host memory held by queued results
                writes.pop(0).result()
        for w in writes:
            w.result()
    return done


class ArrayPool:
    """Recycles the large host arrays of the file drivers. A fresh numpy.empty of half a gigabyte is page-faulted in
    by the device-to-host copy that fills it (about 30 ms per 16-file batch, more than the copy and the kernels
    together); an array whose pages are already mapped is not. An owner array is free again when nothing but the
    pool refers to it: every view handed out (and every view of a view) holds a reference to its owner."""

    def __init__(self, keep=6):
        self._owners = []
        self._lock = threading.Lock()
        self.keep = keep

    def empty(self, count, dtype=numpy.float64):
        """Uninitialised 1-D array of `count` elements backed by a pooled buffer."""
        dtype = numpy.dtype(dtype)
        nbytes = int(count) * dtype.itemsize
        with self._lock:
            best = None
            for i in range(len(self._owners)):
                o = self._owners[i]
                # references: the list, `o`, getrefcount's argument
                if o.nbytes >= nbytes and sys.getrefcount(o) <= 3 and (best is None or o.nbytes < best.nbytes):
                    best = o
                del o
            if best is None:
                best = numpy.empty(max(nbytes + nbytes // 8, 1 << 20), numpy.uint8)
                self._owners.append(best)
                if len(self._owners) > self.keep:   # drop the smallest buffer nobody uses
                    idle = [o for o in self._owners if o is not best and sys.getrefcount(o) <= 4]
                    if idle:
                        victim = min(idle, key=lambda o: o.nbytes)
                        self._owners = [o for o in self._owners if o is not victim]
            return best[:nbytes].view(dtype)


host_pool = ArrayPool()
