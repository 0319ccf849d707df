"""K4 with conv3 + conv4 on the bf16 matrix cores (option cnn_bf16x3) against the float32 kernels and the oracle CNN, and
the time of the CNN stage either way. Diagnostic, GPU box only."""
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
m = F2CNNModel.glorot(7)
rng = np.random.default_rng(11)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14240
x = rng.uniform(0.0, 1.0, size=(n, 11, 128)).astype(np.float32)
res = {}
for opt in (0, 1, 0, 1):
    ctx.set_option("cnn_bf16x3", opt)
    m.predict(x[:256], ctx)
    ctx.prof_enable(True)
    for _ in range(3):
        s = m.predict(x, ctx)
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    cnt, ms = prof["k_cnn"] if "k_cnn" in prof else list(prof.values())[-1]
    res[opt] = s
    print(f"cnn_bf16x3={opt}: CNN stage {ms / cnt:.3f} ms per {n} windows ({prof})", flush=True)
ref = orc.cnn_forward(x[:512], orc.glorot_weights(7))
print("f32 kernels vs oracle (512 windows): max |d score|", float(np.abs(res[0][:512] - ref).max()))
print("bf16x3 conv3+conv4 vs oracle        : max |d score|", float(np.abs(res[1][:512] - ref).max()))
print("bf16x3 vs f32 kernels, all windows  : max |d score|", float(np.abs(res[1] - res[0]).max()),
      " labels differing:", int(((res[1][:, 1] > res[1][:, 0]) != (res[0][:, 1] > res[0][:, 0])).sum()))
