"""Randomised check of the default (split-fp16; split-bf16 in rounds 3-4) CNN path against the oracle CNN (float32, Training.py:93-114): several
weight seeds, input shapes and input statistics. Diagnostic, GPU box only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import f2cnn_oracle as orc
from f2cnn_amd import _lib
from f2cnn_amd.model import F2CNNModel
ctx = _lib.default_context()
rng = np.random.default_rng(2)
worst = 0.0
flips = 0
total = 0
for seed in range(1, 9):
    for rows, channels in ((11, 128), (11, 64), (13, 40), (10, 100)):
        m = F2CNNModel.glorot(seed, rows, channels, zero_bias=bool(seed & 1))
        n = 300
        kind = seed % 3
        x = rng.random((n, rows, channels)).astype(np.float32)
        if kind == 1:
            x = x ** 4                                   # mostly small values
        elif kind == 2:
            x[:, :, ::2] = 0.0                           # structured zeros
        s = m.predict(x, ctx)
        ref = orc.cnn_forward(x, {k: v for k, v in m.tensors.items()})
        d = float(np.abs(s - ref).max())
        worst = max(worst, d)
        clear = np.abs(ref[:, 1] - ref[:, 0]) > 1e-5
        f = int(((s[:, 1] > s[:, 0]) != (ref[:, 1] > ref[:, 0]))[clear].sum())
        flips += f
        total += int(clear.sum())
        print(f"seed {seed} {rows}x{channels} kind {kind}: max |d score| {d:.2e}, labels differing among {int(clear.sum())} clear windows: {f}", flush=True)
print(f"worst {worst:.3e}; {flips} label differences among {total} windows with an oracle margin above 1e-5")
