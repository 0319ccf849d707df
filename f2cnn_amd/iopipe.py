"""File I/O pipeline around the GPU kernels.

The reference hides disk latency behind one process per core (GammatoneFiltering.py:122-125,
EnvelopeExtraction.py:145-149); here one process drives one GPU, so the reads of the next batch and the `.npy`
writes of the previous one run on threads (NumPy file I/O releases the GIL) while the GPU works on the current
batch. Results are written by the caller's `save` exactly as the reference names and formats them.
"""
import os
import threading
import weakref
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
    together); a buffer whose pages are already mapped is not.

    Every array handed out is a lease on one pooled buffer. The lease is tracked explicitly: the array is built over
    a memoryview of the buffer, so every view derived from it (slices, reshapes, the per-file matrices the writer
    threads hold) keeps THAT array alive through its `.base`, and a weakref.finalize on it returns the buffer to the
    pool when the last of them is gone. `alloc(nbytes)` may supply page-locked buffers (any writable object with the
    buffer protocol)."""

    def __init__(self, keep=6, alloc=None):
        self._buffers = []          # [buffer, nbytes, busy]
        self._lock = threading.Lock()
        self._alloc = alloc or (lambda nbytes: numpy.empty(nbytes, numpy.uint8))
        self.keep = keep

    def _release(self, entry):
        with self._lock:
            entry[2] = False
            idle = [e for e in self._buffers if not e[2]]
            if len(self._buffers) > self.keep and len(idle) > 1:   # drop the smallest buffer nobody uses
                victim = min(idle, key=lambda e: e[1])
                self._buffers = [e for e in self._buffers if e is not victim]

    def empty(self, count, dtype=numpy.float64):
        """Uninitialised 1-D array of `count` elements backed by a pooled buffer."""
        dtype = numpy.dtype(dtype)
        nbytes = int(count) * dtype.itemsize
        with self._lock:
            fits = [e for e in self._buffers if not e[2] and e[1] >= nbytes]
            if fits:
                entry = min(fits, key=lambda e: e[1])
            else:
                size = max(nbytes + nbytes // 8, 1 << 20)
                entry = [self._alloc(size), size, False]
                self._buffers.append(entry)
            entry[2] = True
        arr = numpy.frombuffer(memoryview(entry[0]).cast("B")[:nbytes], dtype=dtype)
        weakref.finalize(arr, self._release, entry)
        return arr


host_pool = ArrayPool()
