"""Where the time of `prepare filter` goes: wraps load / compute / collect / save of run_batches with timers."""
import contextlib, io, os, sys, time, tempfile, shutil, threading
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from f2cnn_amd import cli, config, wavio, _lib, iopipe
n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cmd = sys.argv[2:] or ["prepare", "filter"]
base = tempfile.mkdtemp(prefix="f2tl_", dir="/dev/shm"); os.chdir(base)
config.write_default(); os.makedirs("resources/f2cnn/TEST")
waves = bench.synth_batch(2029, 0, n_files, 16000)
for i in range(n_files):
    wavio.write_sphere(f"resources/f2cnn/TEST/DR1.S{i:04d}.SA1.WAV", 16000, waves[i])
_lib.default_context()
def quiet(argv):
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(argv)
quiet(["prepare", "features", "--cutoff", "50"])
orig = iopipe.run_batches
acc = {"load": 0.0, "compute": 0.0, "collect": 0.0, "save": 0.0}
lock = threading.Lock()
def timed_run(items, load, compute, save, **kw):
    def wrap(name, fn):
        def inner(*a):
            t = time.perf_counter(); r = fn(*a); dt = time.perf_counter() - t
            with lock: acc[name] += dt
            return r
        return inner
    def compute2(*a):
        t = time.perf_counter(); r = compute(*a); acc["compute"] += time.perf_counter() - t
        if callable(r):
            def fin():
                t = time.perf_counter(); out = list(r()); acc["collect"] += time.perf_counter() - t
                return out
            return fin
        return r
    return orig(items, wrap("load", load), compute2, wrap("save", save), **kw)
import f2cnn_amd.scripts.processing.GammatoneFiltering as G, f2cnn_amd.scripts.processing.EnvelopeExtraction as E
G.run_batches = E.run_batches = timed_run
t = time.perf_counter(); quiet(cmd); wall = time.perf_counter() - t
print(cmd, "wall %.3f s" % wall, {k: round(v, 3) for k, v in acc.items()}, "(load/save are summed over threads)")
shutil.rmtree(base, ignore_errors=True)
