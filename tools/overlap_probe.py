"""Do K1 (float64-FMA bound) and K2 (HBM/LDS bound) overlap when they run on two streams?
Two contexts = two streams on one device. Times R rounds of [K1 on A] then [K2 on B] back to back on one stream,
and the same launches issued to both streams at once (K1 of batch i+1 beside K2 of batch i)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
B, C, N, R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 128, 16000, 10
a, b = _lib.Context(0), _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
waves = bench.synth_batch(2027, 0, B, N)
off = np.arange(B + 1, dtype=np.int64) * N
d_wave = a.malloc(waves.nbytes); a.h2d(d_wave, waves)
d_g0, d_g1, d_e = a.malloc(8 * C * N * B), a.malloc(8 * C * N * B), a.malloc(8 * C * N * B)
k1 = lambda ctx, dst: ctx.erb_filterbank_batch(d_wave, 0, off, coefs, B, C, dst, 1)
k2 = lambda ctx, src: ctx.envelope_batch(src, off, B, C, True, 50.0, 0, d_e, 1)
k1(a, d_g0); k1(a, d_g1); k2(b, d_g0); a.synchronize(); b.synchronize()
def wall(fn):
    a.synchronize(); b.synchronize(); t = time.perf_counter(); fn(); a.synchronize(); b.synchronize()
    return (time.perf_counter() - t) / R * 1e3
def serial():
    for i in range(R):
        k1(a, d_g1); k2(a, d_g0)
def only1():
    for i in range(R): k1(a, d_g1)
def only2():
    for i in range(R): k2(b, d_g0)
def both():
    for i in range(R):
        k1(a, d_g1); k2(b, d_g0)      # independent buffers: K1 writes g1 while K2 reads g0
for name, fn in (("K1 alone (f64 out)", only1), ("K2 alone (f64 in)", only2), ("one stream K1;K2", serial), ("two streams K1||K2", both)):
    fn()
    print(f"{name:24s} {wall(fn):7.3f} ms per round", flush=True)
