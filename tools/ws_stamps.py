"""Phase split of the weight-stationary CNN kernels (f2_cnn_ws.hip): one forward pass over n windows with the -DF2_WS_STAMPS
library (tools/build_variant.sh wsstamps -DF2_WS_STAMPS); the library prints the per-wave means to stderr. Never used for timing."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f2cnn_amd import build
build.LIB_PATH = os.path.abspath(os.environ.get("F2CNN_PROBE_LIB", "tools/libf2cnn_hip_wsstamps.so"))
from f2cnn_amd import _lib
from f2cnn_amd.model import F2CNNModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14240
ctx = _lib.Context(0)
m = F2CNNModel.glorot(7)
h = m.handle(ctx)
x = np.random.default_rng(0).uniform(0, 1, (n, 11, 128)).astype(np.float32)
d_x = ctx.malloc(x.nbytes); ctx.h2d(d_x, x)
d_s = ctx.malloc(8 * n); d_l = ctx.malloc(n)
for _ in range(3):
    ctx.cnn_forward(h, d_x, n, d_s, d_l, _lib.MEM_DEVICE)
ctx.synchronize()
