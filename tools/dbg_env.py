import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import f2cnn_oracle as orc
from f2cnn_amd import _lib
ctx = _lib.default_context()
def run(N, inplace, zero_tail, lpf=False):
    m = np.random.default_rng(5).standard_normal((1, N))
    ref = orc.extract_envelope_from_matrix(m, lpf, 50)
    big = np.full(40000, 1e6); big[:N] = m[0]
    if zero_tail: big[N:] = 0
    d_in = ctx.malloc(big.nbytes); ctx.h2d(d_in, big)
    d_out = d_in if inplace else ctx.malloc(big.nbytes)
    if not inplace: ctx.h2d(d_out, np.full(40000, -7.0))
    off = np.array([0, N], np.int64)
    ctx.envelope_batch(d_in, off, 1, 1, lpf, 50.0, _lib.FFT_F32, d_out, _lib.MEM_DEVICE)
    out = np.empty(40000); ctx.d2h(out, d_out)
    err = np.abs(out[:N] - ref[0])
    print(N, "inplace", inplace, "zero_tail", zero_tail, "maxerr", err.max(), "tail untouched", np.array_equal(out[N:], (big if inplace else np.full(40000,-7.0))[N:]))
for N in (30000,):
    for inplace in (False, True):
        for zt in (False, True):
            run(N, inplace, zt)
run(16000, True, False); run(16000, False, False)
