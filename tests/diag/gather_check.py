"""Every-sample windows: blocked two-pass gather (option gather_blocked, default) against the one-workgroup-per-window kernel
- must be bit-identical - and the time of either. Diagnostic, GPU box only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from f2cnn_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(5)
for C, N in ((128, 16000), (64, 9000), (40, 5000), (190, 4000)):
    env = np.abs(rng.standard_normal((C, N))) + 1e-3
    nb = N - 11 * 160
    outs = {}
    for opt in (1, 0):
        ctx.set_option("gather_blocked", opt)
        out = np.empty((nb, 11, C), np.float32)
        ctx.gather_windows(env, C, N, None, nb, 5, 160, True, out, _lib.MEM_HOST)
        outs[opt] = out
    same = np.array_equal(outs[0], outs[1])
    print(f"C={C} N={N}: {nb} windows, blocked == per-window: {same}; max |diff| {np.abs(outs[0] - outs[1]).max():.2e}", flush=True)
ctx.set_option("gather_blocked", 1)
