"""Which windows differ between the weight-stationary CNN kernels and the per-tile ones, for window counts that give a workgroup
only a few tiles. Diagnostic, GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from f2cnn_amd import _lib
from f2cnn_amd.model import F2CNNModel
ctx = _lib.default_context()
for rows, ch in ((11, 128), (11, 64), (10, 100)):
    m = F2CNNModel.glorot(7, rows, ch, zero_bias=False)
    for n in (1, 3, 63, 64, 65, 128, 129, 257, 300, 600, 1025):
        x = np.random.default_rng(3).random((n, rows, ch)).astype(np.float32)
        out = {}
        for ws in (0, 1):
            ctx.set_option("cnn_ws", ws)
            out[ws] = m.predict(x, ctx)
        ctx.set_option("cnn_ws", 1)
        bad = np.flatnonzero(np.abs(out[1] - out[0]).max(axis=1) > 1e-5)
        print(rows, ch, n, "bad windows:", len(bad), bad[:12], bad[-6:] if len(bad) else "", flush=True)
