"""File I/O pipeline around the GPU kernels.

The reference hides disk latency behind one process per core (GammatoneFiltering.py:122-125,
EnvelopeExtraction.py:145-149); here one process drives one GPU, so the reads of the next batch and the `.npy`
writes of the previous one run on threads (NumPy file I/O releases the GIL) while the GPU works on the current
batch. Results are written by the caller's `save` exactly as the reference names and formats them.
"""
import json as _json
import os
import threading
import time as _time
import weakref
from concurrent.futures import ThreadPoolExecutor

import numpy


class Unreadable:
    """What a guarded `load` returns for an input that could not be read (the reason travels with it)."""

    def __init__(self, reason):
        self.reason = reason


class JobReport:
    """Bookkeeping of one file-level command (the reference only prints; here a failing file does not
    end the run, a re-run can skip what is already done, and the totals can be written as JSON).

      skip_existing   inputs whose outputs already exist and are not older than the input are left alone (resume)
      failed          inputs that could not be read or processed: reported, skipped, and the command's exit status
                      becomes 2 (the reference would have died in the worker, GammatoneFiltering.py:69-83)
      metrics         path of a JSON file: command, rank/world, files, skipped, failed, audio seconds, wall seconds,
                      audio-seconds per second
    """

    def __init__(self, command, skip_existing=False, metrics=None):
        from .runtime import rank_world
        self.command, self.skip_existing, self.metrics = command, bool(skip_existing), metrics
        self.rank, self.world = rank_world()
        self.started = _time.time()
        self.done = self.skipped = 0
        self.failed = []
        self.audio_s = 0.0
        self._lock = threading.Lock()

    def pending(self, items, outputs_of, params=None, stamp_dir=None):
        """The items still to do: all of them, or with skip_existing those with a missing, stale or incomplete output
        (an .npy whose header promises more bytes than the file holds - what a killed run leaves behind if it did not
        write through save_npy_atomic - counts as missing).

        params / stamp_dir: the settings the outputs depend on (low-pass on / cutoff, ...) live in ONE stamp per command,
        `<stamp_dir>/.f2cnn_<command>.json` = {"params": ..., "since": time}. The stamp is rewritten only when the settings
        change, with `since` = now, and from then on an output older than `since` is stale whatever run left it: a run
        with new settings that dies part-way, resumed, recomputes exactly the outputs the old settings made. (Outputs
        found without any stamp are trusted, `since` = 0. Every rank reads / writes the same file: a resume with another
        world size sees the same stamp.)"""
        since = 0.0
        if params is not None and stamp_dir is not None:
            stamp = os.path.join(stamp_dir, ".f2cnn_{}.json".format(self.command.replace(" ", "_")))
            previous = None
            try:
                with open(stamp) as f:
                    previous = _json.load(f)
                if not (isinstance(previous, dict) and "params" in previous and "since" in previous):
                    previous = None
            except (OSError, ValueError):
                pass
            if previous is not None and previous["params"] == params:
                since = float(previous["since"])
            else:
                since = _time.time() if previous is not None else 0.0
                try:
                    tmp = "{}.{}.tmp".format(stamp, os.getpid())
                    with open(tmp, "w") as f:
                        _json.dump({"params": params, "since": since}, f)
                    os.replace(tmp, stamp)
                except OSError:
                    pass
                if self.skip_existing and previous is not None:
                    print("Settings changed since the outputs were written ({} -> {}): nothing is skipped.".format(
                        previous["params"], params))
        if not self.skip_existing:
            return list(items)
        todo = []
        for it in items:
            outs = outputs_of(it)
            fresh = all(os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(it), since) and npy_complete(o)
                        for o in outs)
            if fresh:
                self.skipped += 1
            else:
                todo.append(it)
        if self.skipped:
            print("{} of {} files are up to date and skipped.".format(self.skipped, len(items)))
        return todo

    def guard(self, load):
        """load(item, ...) that returns Unreadable(reason) instead of raising."""
        def guarded(item, *rest):
            try:
                return load(item, *rest)
            except Exception as exc:      # a corrupt or unsupported file: the other files of the corpus still run
                self.fail(item, exc)
                return Unreadable(repr(exc))
        return guarded

    def fail(self, item, exc):
        with self._lock:
            self.failed.append((str(item), repr(exc)))
        print("ERROR: {} skipped: {}".format(item, exc))

    def add(self, samples, framerate):
        with self._lock:
            self.done += 1
            self.audio_s += samples / float(framerate)
            return self.done

    def finish(self):
        wall = _time.time() - self.started
        out = {"command": self.command, "rank": self.rank, "world": self.world, "files": self.done,
               "files_skipped": self.skipped, "files_failed": len(self.failed), "failed": self.failed,
               "audio_seconds": round(self.audio_s, 3), "wall_s": round(wall, 3),
               "audio_s_per_s": round(self.audio_s / wall, 2) if wall > 0 else None}
        if self.metrics:
            path = self.metrics if self.world == 1 else "{}.rank{}".format(self.metrics, self.rank)
            with open(path, "w") as f:
                _json.dump(out, f)
        if self.failed:
            print("{} file(s) could not be processed.".format(len(self.failed)))
        return out

    @property
    def exit_status(self):
        return 2 if self.failed else 0


def _workers(kind):
    """Thread counts for the box's CPU share (16 cores for one GPU): numpy.save to tmpfs reaches 25 GB/s with 8
    threads and 43 GB/s with 16, reads into pooled buffers 41 / 61 GB/s (tools/io_probe.py)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    cores = max(2, min(cores, 16))
    return max(2, cores - 6) if kind == "read" else max(4, cores - 4)


def run_batches(items, load, compute, save, batch=32, readers=None, writers=None, max_pending_writes=64, plan=None,
                depth=2):
    """The file pipeline of the drivers:

        plan(chunk) -> (batch_state, slots)   optional, main thread: e.g. read the headers, take one pooled buffer for
                                              the batch and hand every item its slice of it
        load(item[, slot])                    reader threads (NumPy / file I/O releases the GIL)
        compute(loaded[, batch_state])        main thread; loaded = [(item, what load returned)]. Returns an iterable
                                              of (item, result) or, for asynchronous device work, a zero-argument
                                              callable that waits for the work and returns that iterable
        save(item, result)                    writer threads

    Up to `depth` computes stay un-collected (the drivers alternate two device contexts, so one batch's copies run
    beside the next batch's kernels), the loads of the next batches run ahead on the reader threads, and the writes of
    earlier batches drain on the writer threads. Order of `compute` calls follows `items`; returns the number of items
    saved."""
    items = list(items)
    if not items:
        return 0
    readers = readers or _workers("read")
    writers = writers or _workers("write")
    done = 0
    with ThreadPoolExecutor(readers) as rpool, ThreadPoolExecutor(writers) as wpool:
        def submit_loads(chunk):
            if plan is None:
                return None, [(it, rpool.submit(load, it)) for it in chunk]
            state, slots = plan(chunk)
            return state, [(it, rpool.submit(load, it, slot)) for it, slot in zip(chunk, slots)]
        chunks = [items[s:s + batch] for s in range(0, len(items), batch)]
        ahead = [submit_loads(chunks[0])]
        if len(chunks) > 1:
            ahead.append(submit_loads(chunks[1]))
        in_flight, writes = [], []

        def collect(res):
            nonlocal done
            for it, out in (res() if callable(res) else res):
                writes.append(wpool.submit(save, it, out))
                done += 1
            while len(writes) > max_pending_writes:     # bound the host memory held by queued results
                writes.pop(0).result()
        for ci in range(len(chunks)):
            state, pending = ahead.pop(0)
            loaded = [(it, f.result()) for it, f in pending]
            if ci + 2 < len(chunks):
                ahead.append(submit_loads(chunks[ci + 2]))
            in_flight.append(compute(loaded) if plan is None else compute(loaded, state))
            while len(in_flight) >= max(depth, 1):
                collect(in_flight.pop(0))
        while in_flight:
            collect(in_flight.pop(0))
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
        # re-entrant: a lease's finalizer (-> _release) may run inside empty() when a garbage collection starts there
        self._lock = threading.RLock()
        self._alloc = alloc or (lambda nbytes: numpy.empty(nbytes, numpy.uint8))
        self._retired = []          # buffers dropped from the pool, freed by the thread that calls empty()
        self.keep = keep

    def _release(self, entry):
        # may run on a writer thread or inside a collection: only bookkeeping here. A buffer that leaves the pool is
        # parked in _retired and dropped by the next empty() call - i.e. on the driver's thread, which owns the device
        # context the page-locked buffers were allocated from (a context is not thread-safe)
        with self._lock:
            entry[2] = False
            idle = [e for e in self._buffers if not e[2]]
            if len(self._buffers) > self.keep and len(idle) > 1:   # drop the smallest buffer nobody uses
                victim = min(idle, key=lambda e: e[1])
                self._buffers = [e for e in self._buffers if e is not victim]
                self._retired.append(victim[0])

    def empty(self, count, dtype=numpy.float64):
        """Uninitialised 1-D array of `count` elements backed by a pooled buffer."""
        dtype = numpy.dtype(dtype)
        nbytes = int(count) * dtype.itemsize
        with self._lock:
            retired, self._retired = self._retired, []
            fits = [e for e in self._buffers if not e[2] and e[1] >= nbytes]
            entry = min(fits, key=lambda e: e[1]) if fits else None
            if entry is not None:
                entry[2] = True
        del retired                  # (frees page-locked memory here, outside the lock, on the calling thread)
        if entry is None:
            size = max(nbytes + nbytes // 8, 1 << 20)
            entry = [self._alloc(size), size, True]     # allocated outside the lock
            with self._lock:
                self._buffers.append(entry)
        arr = numpy.frombuffer(memoryview(entry[0]).cast("B")[:nbytes], dtype=dtype)
        weakref.finalize(arr, self._release, entry)
        return arr


def _pinned(nbytes):
    """Page-locked buffer from the default device context (hipHostMalloc through f2_host_alloc): copies to and from
    it can be queued asynchronously (F2_MEM_HOST_ASYNC)."""
    from . import _lib
    return _lib.default_context().host_alloc(nbytes)


host_pool = ArrayPool(keep=int(os.environ.get("F2CNN_POOL_KEEP", "16")), alloc=_pinned)


def save_npy_atomic(filename, array):
    """numpy.save(filename, array) that never leaves a partial file under the final name: the array goes to
    `<name>.tmp.<pid>` in the same directory and is renamed over the target (os.replace) once it is complete.
    `filename` gets the `.npy` suffix numpy.save would add.

    For single outputs. The batch writers of `prepare filter / envelope / features` do NOT use it: twelve writer threads
    creating and renaming 16 MB files in one tmpfs directory ran the whole pipeline at 0.4x the rate of in-place
    numpy.save (measured, same box: 290-360 against 680-990 audio-s/s). Their protection against a killed run is
    JobReport.pending(): with --skip-existing an output only counts as done when its NPY header fits the file size
    (npy_complete), which a partially written file never satisfies."""
    final = filename if str(filename).endswith(".npy") else str(filename) + ".npy"
    tmp = "{}.tmp.{}".format(final, os.getpid())
    try:
        with open(tmp, "wb") as f:
            numpy.save(f, array)
        os.replace(tmp, final)
    except BaseException:
        try:
            os.remove(tmp)
        except OSError:
            pass
        raise
    return final


def npy_complete(path):
    """True unless `path` is an .npy file whose header describes more payload than the file contains."""
    if not str(path).endswith(".npy"):
        return True
    try:
        with open(path, "rb") as f:
            version = numpy.lib.format.read_magic(f)
            reader = numpy.lib.format.read_array_header_1_0 if version == (1, 0) else numpy.lib.format.read_array_header_2_0
            shape, _, dtype = reader(f)
            need = f.tell() + int(numpy.prod(shape, dtype=numpy.int64)) * dtype.itemsize
        return os.path.getsize(path) >= need
    except (OSError, ValueError):
        return False


def npy_layout(path):
    """(shape, data offset) of an NPY file holding a C-order '<f8' array, or None for anything else."""
    with open(path, "rb") as f:
        try:
            version = numpy.lib.format.read_magic(f)
            reader = numpy.lib.format.read_array_header_1_0 if version == (1, 0) else numpy.lib.format.read_array_header_2_0
            shape, fortran, dtype = reader(f)
        except ValueError:
            return None
        if fortran or dtype != numpy.dtype('<f8'):
            return None
        return shape, f.tell()


def read_npy_into(path, offset, dest):
    """Payload of an NPY file (see npy_layout) straight into `dest` (a writable contiguous array of the right size)."""
    view = memoryview(dest).cast("B")
    with open(path, "rb", buffering=0) as f:
        f.seek(offset)
        got = 0
        while got < len(view):
            n = f.readinto(view[got:])
            if not n:
                raise ValueError("{}: truncated NPY payload".format(path))
            got += n
    return dest
