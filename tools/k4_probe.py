"""K4 probe: the CNN forward pass on 4096 device-resident windows, 10 launches. Run under
`rocprofv3 --kernel-trace --stats` for per-layer times; F2CNN_PROBE_LIB picks a tools/build_variant.sh library."""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
if os.environ.get("F2CNN_PROBE_LIB"):
    from f2cnn_amd import build
    build.LIB_PATH = os.path.abspath(os.environ["F2CNN_PROBE_LIB"])
from f2cnn_amd import _lib
from f2cnn_amd.model import F2CNNModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.Context(0)
m = F2CNNModel.glorot(7)
h = m.handle(ctx)
x = np.random.default_rng(0).uniform(0, 1, (n, 11, 128)).astype(np.float32)
d_x = ctx.malloc(x.nbytes); ctx.h2d(d_x, x)
d_s = ctx.malloc(8 * n); d_l = ctx.malloc(n)
ctx.cnn_forward(h, d_x, n, d_s, d_l, _lib.MEM_DEVICE); ctx.synchronize()
ctx.prof_enable(True)
for _ in range(10):
    ctx.cnn_forward(h, d_x, n, d_s, d_l, _lib.MEM_DEVICE)
p = ctx.prof_get()
for k, (c, t) in p.items():
    print(k, f"{t / 10:.3f} ms per forward of {n} windows = {n * 41.98e6 / (t / 10 * 1e-3) / 1e12:.1f} TFLOP/s", flush=True)
