"""Host I/O building blocks of the file drivers on this box: copy rates (pageable vs page-locked) and
numpy.save / numpy.load rates on tmpfs by thread count."""
import os, sys, time, tempfile, shutil, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, "/root/repo")
from f2cnn_amd import _lib
ctx = _lib.default_context()
MB = 1 << 20
n = 512 * MB
d = ctx.malloc(n)
page = np.empty(n, np.uint8); page[:] = 1
pin = np.frombuffer(ctx.host_alloc(n), np.uint8); pin[:] = 1
def rate(fn, reps=4):
    fn(); t = time.perf_counter()
    for _ in range(reps): fn()
    return n * reps / (time.perf_counter() - t) / 1e9
print(f"H2D pageable {rate(lambda: ctx.h2d(d, page)):.1f} GB/s   pinned {rate(lambda: ctx.h2d(d, pin)):.1f} GB/s")
print(f"D2H pageable {rate(lambda: ctx.d2h(page, d)):.1f} GB/s   pinned {rate(lambda: ctx.d2h(pin, d)):.1f} GB/s", flush=True)
base = tempfile.mkdtemp(prefix="f2io_", dir="/dev/shm")
mat = np.random.default_rng(0).random((128, 16000))
files = 96
for src_name, src in (("pageable", mat), ("pinned", np.frombuffer(ctx.host_alloc(mat.nbytes), np.float64).reshape(128, 16000))):
    src[...] = mat
    for th in (1, 4, 8, 12, 16):
        with ThreadPoolExecutor(th) as ex:
            t = time.perf_counter()
            list(ex.map(lambda i: np.save(f"{base}/f{i}.npy", src), range(files)))
            dt = time.perf_counter() - t
        print(f"np.save from {src_name} x{th:2d} threads: {files * mat.nbytes / dt / 1e9:6.2f} GB/s", flush=True)
for th in (1, 4, 8, 12, 16):
    with ThreadPoolExecutor(th) as ex:
        t = time.perf_counter()
        list(ex.map(lambda i: np.load(f"{base}/f{i}.npy"), range(files)))
        dt = time.perf_counter() - t
    print(f"np.load x{th:2d} threads: {files * mat.nbytes / dt / 1e9:6.2f} GB/s", flush=True)
def readinto(i, dst):
    with open(f"{base}/f{i}.npy", "rb") as f:
        np.lib.format.read_magic(f); np.lib.format.read_array_header_1_0(f)
        f.readinto(memoryview(dst).cast("B"))
bufs = [np.frombuffer(ctx.host_alloc(mat.nbytes), np.float64) for _ in range(16)]
for th in (4, 8, 16):
    with ThreadPoolExecutor(th) as ex:
        t = time.perf_counter()
        list(ex.map(lambda i: readinto(i, bufs[i % 16]), range(files)))
        dt = time.perf_counter() - t
    print(f"readinto pinned x{th:2d} threads: {files * mat.nbytes / dt / 1e9:6.2f} GB/s", flush=True)
shutil.rmtree(base)
print("cpus", len(os.sched_getaffinity(0)), os.cpu_count())
