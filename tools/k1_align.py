"""K1 against the alignment of the output rows (diagnostic): utterance lengths around 16000 through the fused call
(float32 hand-off) and the float64 filterbank output."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import os
if os.environ.get("F2CNN_PROBE_LIB"):   # a tools/build_variant.sh library
    from f2cnn_amd import build
    build.LIB_PATH = os.path.abspath(os.environ["F2CNN_PROBE_LIB"])
from f2cnn_amd import _lib
from f2cnn_amd.gammatone import filters
import bench
C = 128
ctx = _lib.Context(0)
coefs = filters.make_erb_filters(16000, filters.centre_freqs(16000, C, 100))
for n in ([int(a) for a in sys.argv[1:]] or (16000, 15999, 16002, 16008, 16016, 16001)):
    B = 1000
    waves = bench.synth_batch(7, 0, B, n)
    off = np.arange(B + 1, dtype=np.int64) * n
    d_wave = ctx.malloc(waves.nbytes); ctx.h2d(d_wave, waves)
    d_out = ctx.malloc(8 * C * n * B)
    for name, run in (("fused f32 hand-off", lambda: ctx.filterbank_envelope_fused(d_wave, 0, off, coefs, B, C, True, 50.0, 0, d_out, None, 1)),
                      ("filterbank f64 out", lambda: ctx.erb_filterbank_batch(d_wave, 0, off, coefs, B, C, d_out, 1))):
        run(); ctx.synchronize(); ctx.prof_enable(True)
        for _ in range(5): run()
        p = ctx.prof_get(); ctx.prof_enable(False)
        print(n, name, {k: round(t / c, 3) for k, (c, t) in p.items()}, flush=True)
    ctx.free(d_wave); ctx.free(d_out)
