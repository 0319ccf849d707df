"""Ablation probe for K2 on device buffers: times f2_envelope_batch / fused in a few modes."""
import sys, time, numpy as np
import os
sys.path.insert(0, "/root/repo")
if os.environ.get("F2CNN_PROBE_LIB"):   # a tools/build_variant.sh library
    from f2cnn_amd import build
    build.LIB_PATH = os.path.abspath(os.environ["F2CNN_PROBE_LIB"])
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
B, C, N = int(sys.argv[1]) if len(sys.argv) > 1 else 500, 128, 16000
ctx = _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
waves = bench.synth_batch(2027, 0, B, N)
off = np.arange(B + 1, dtype=np.int64) * N
d_wave = ctx.malloc(waves.nbytes); ctx.h2d(d_wave, waves)
d_a = ctx.malloc(8 * C * N * B); d_b = ctx.malloc(8 * C * N * B)
def timeit(name, fn, reps=5):
    fn(); ctx.synchronize(); ctx.prof_enable(True)
    for _ in range(reps): fn()
    p = ctx.prof_get(); ctx.prof_enable(False)
    print(name, {k: round(ms / n, 3) for k, (n, ms) in p.items()}, flush=True)
ctx.erb_filterbank_batch(d_wave, 0, off, coefs, B, C, d_a, 1); ctx.synchronize()
timeit("env f64-in lpf50 out-of-place", lambda: ctx.envelope_batch(d_a, off, B, C, True, 50.0, 0, d_b, 1))
timeit("env f64-in nolpf  out-of-place", lambda: ctx.envelope_batch(d_a, off, B, C, False, 0.0, 0, d_b, 1))
timeit("fused lpf50", lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 0, d_b, None, 1))
timeit("fused nolpf", lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, False, 0.0, 0, d_b, None, 1))
timeit("fused lpf50 f64fft", lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 1, d_b, None, 1))
# correctness of whatever library is loaded: three rows of the fused output against the oracle
sys.path.insert(0, "/root/repo/oracle")
import f2cnn_oracle as orc
for lpf in (True, False):
    ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, lpf, 50.0, 0, d_b, None, 1); ctx.synchronize()
    worst = 0.0
    for b, c in ((0, 0), (B // 2, 64), (B - 1, 127)):
        got = np.empty(N); ctx.d2h(got, d_b + 8 * ((b * C + c) * N))
        ref = orc.extract_envelope_from_matrix(orc.erb_filterbank(waves[b], coefs[c:c + 1]), lpf, 50)[0]
        worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
    print("check fused lpf" if lpf else "check fused nolpf", "max rel err", worst, flush=True)
