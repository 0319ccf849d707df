#!/usr/bin/env python3
"""End-to-end file benchmark (with disk I/O): N synthetic 1 s SPHERE files -> .GFB.npy + .ENV1.npy.
Reference figure for the same job (BASELINE.md section 2, 8 CPU cores, 64 files): 9.4 audio-s/s."""
import contextlib, io, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from f2cnn_amd import cli, config, wavio, _lib

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 128
base = tempfile.mkdtemp(prefix="f2bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
os.chdir(base)
config.write_default()
os.makedirs("resources/f2cnn/TEST")
waves = bench.synth_batch(2029, 0, n_files, 16000)
for i in range(n_files):
    wavio.write_sphere(f"resources/f2cnn/TEST/DR1.S{i:04d}.SA1.WAV", 16000, waves[i])
_lib.default_context()                      # context + library load outside the timed region
def run(argv):
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        assert cli.main(argv) == 0
    return time.perf_counter() - t
run(["prepare", "features", "--cutoff", "50"])          # warm-up (twiddles, allocations, page cache)
res = {"files": n_files, "audio_s": n_files}
t1 = run(["prepare", "filter"]); t2 = run(["prepare", "envelope", "--cutoff", "50"])
res["filter_s"], res["envelope_s"] = round(t1, 3), round(t2, 3)
res["two_commands_audio_s_per_s"] = round(n_files / (t1 + t2), 1)
t3 = run(["prepare", "features", "--cutoff", "50"])
res["features_s"] = round(t3, 3)
res["one_pass_audio_s_per_s"] = round(n_files / t3, 1)
res["bytes_written_per_pass_MB"] = round(n_files * 2 * 128 * 16000 * 8 / 1e6, 1)
print(json.dumps(res))
shutil.rmtree(base, ignore_errors=True)
