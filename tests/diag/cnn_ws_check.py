"""K4: weight-stationary split-bf16 kernels (option cnn_ws, f2_cnn_ws.hip) against the per-tile ones, the float32 kernels and
the oracle CNN, with the time of the CNN stage each way. Diagnostic, GPU box only."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import f2cnn_oracle as orc
if os.environ.get("F2CNN_PROBE_LIB"):
    from f2cnn_amd import build
    build.LIB_PATH = os.path.abspath(os.environ["F2CNN_PROBE_LIB"])
from f2cnn_amd import _lib
from f2cnn_amd.model import F2CNNModel

ctx = _lib.default_context()
m = F2CNNModel.glorot(7, zero_bias=False)
rng = np.random.default_rng(11)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14240
x = rng.uniform(0.0, 1.0, size=(n, 11, 128)).astype(np.float32)
res = {}
for name, bf, ws in (("f32", 0, 0), ("bf16x3", 1, 0), ("ws", 1, 1), ("bf16x3", 1, 0), ("ws", 1, 1)):
    ctx.set_option("cnn_bf16x3", bf)
    ctx.set_option("cnn_ws", ws)
    m.predict(x[:256], ctx)
    ctx.prof_enable(True)
    for _ in range(3):
        s = m.predict(x, ctx)
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    cnt, ms = prof["k_cnn_forward"]
    res[name] = s
    print(f"{name}: CNN stage {ms / cnt:.3f} ms per {n} windows", flush=True)
ref = orc.cnn_forward(x[:512], dict(m.tensors))
for name in ("f32", "bf16x3", "ws"):
    print(f"{name} vs oracle (512 windows): max |d score| {float(np.abs(res[name][:512] - ref).max()):.3e}")
for name in ("bf16x3", "ws"):
    d = res[name] - res["f32"]
    print(f"{name} vs f32 kernels, all windows: max |d score| {float(np.abs(d).max()):.3e}  labels differing:",
          int(((res[name][:, 1] > res[name][:, 0]) != (res["f32"][:, 1] > res["f32"][:, 0])).sum()))
