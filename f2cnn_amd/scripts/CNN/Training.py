"""Hot-path part of the reference's ``scripts/CNN/Training.py``: ``normalizeInput`` (:13-28). The training
loop itself (:47-156) is outside this path; the network it defines (:93-114) lives in ``f2cnn_amd.model``."""
import numpy

from ... import _lib


def normalizeInput(matrix, ctx=None):
    """Per-window log min-max normalisation: (ln x - ln min)/(ln max - ln min); a constant window gives
    zeros; any value <= 0 raises ValueError("values must all be positive"). Same dtype and shape out.
    Runs on the GPU through the window kernel (one window, step 1)."""
    ctx = ctx or _lib.default_context()
    m = numpy.asarray(matrix)
    shape, dtype = m.shape, m.dtype
    if m.ndim < 2:
        raise ValueError("matrix must be at least two dimensional")
    rows = shape[0]
    flat = numpy.ascontiguousarray(m.reshape(rows, -1), dtype=numpy.float64)
    if rows % 2 == 0:   # the kernel takes 2*radius+1 rows; pad an even window with a copy of its last row
        flat = numpy.vstack([flat, flat[-1:]])
    R, Cn = flat.shape
    env = numpy.ascontiguousarray(flat.T)           # (C, N=R): env[c, k]
    out = numpy.empty((1, R, Cn), numpy.float32)
    try:
        ctx.gather_windows(env, Cn, R, numpy.array([R // 2], numpy.int64), 1, R // 2, 1, True, out, _lib.MEM_HOST)
    except _lib.F2Error as e:
        if e.code == _lib.F2_ERR_NONPOSITIVE:
            print(shape)
            raise ValueError("values must all be positive")
        raise
    res = out[0, :rows].reshape(shape)
    return res.astype(dtype) if numpy.issubdtype(dtype, numpy.floating) else res.astype(numpy.float64)
