"""ctypes binding of libf2cnn_hip.so (C ABI: include/f2cnn_hip.h).

There is no CPU fallback: if the library is missing it is built with hipcc; if that fails, or no
gfx950 device is present when a context is requested, an exception is raised.
"""
import ctypes as C
import importlib.util
import os
import sys
import threading

import numpy as np

from . import build as _build

F2_OK, F2_ERR_INVALID, F2_ERR_HIP, F2_ERR_UNSUPPORTED, F2_ERR_NOMEM, F2_ERR_NONPOSITIVE = 0, -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE, MEM_HOST_ASYNC = 0, 1, 2
WAVE_I16, WAVE_F64 = 0, 1
FFT_F32, FFT_F64 = 0, 1
K_COUNT = 7

_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_P = C.POINTER

# name -> (restype, argtypes); mirrors include/f2cnn_hip.h one to one
SIGNATURES = {
    "f2_version": (_i, []),
    "f2_device_count": (_i, [_P(_i)]),
    "f2_ctx_create": (_i, [_i, _P(_vp)]),
    "f2_ctx_destroy": (_i, [_vp]),
    "f2_ctx_synchronize": (_i, [_vp]),
    "f2_ctx_set_stream": (_i, [_vp, _vp]),
    "f2_ctx_get_stream": (_vp, [_vp]),
    "f2_last_error": (C.c_char_p, [_vp]),
    "f2_ctx_set_option": (_i, [_vp, C.c_char_p, _d]),
    "f2_ctx_get_option": (_i, [_vp, C.c_char_p, _P(_d)]),
    "f2_spectral_guard_read": (_i, [_vp, _vp, _i64, _P(_i64)]),
    "f2_dev_malloc": (_i, [_vp, C.c_size_t, _P(_vp)]),
    "f2_dev_free": (_i, [_vp, _vp]),
    "f2_dev_memset": (_i, [_vp, _vp, _i, C.c_size_t]),
    "f2_host_alloc": (_i, [_vp, C.c_size_t, _P(_vp)]),
    "f2_host_free": (_i, [_vp, _vp]),
    "f2_memcpy_h2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "f2_memcpy_d2h": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "f2_event_create": (_i, [_vp, _P(_vp)]),
    "f2_event_destroy": (_i, [_vp, _vp]),
    "f2_event_record": (_i, [_vp, _vp]),
    "f2_event_elapsed_ms": (_i, [_vp, _vp, _vp, _P(C.c_float)]),
    "f2_event_query": (_i, [_vp, _vp, _P(C.c_int)]),
    "f2_prof_enable": (_i, [_vp, _i]),
    "f2_prof_reset": (_i, [_vp]),
    "f2_prof_get": (_i, [_vp, _i, _P(_i), _P(C.c_float)]),
    "f2_prof_kernel_name": (C.c_char_p, [_i]),
    "f2_erb_filterbank_batch": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _i]),
    "f2_envelope_batch": (_i, [_vp, _vp, _vp, _i, _i, _i, _d, _i, _vp, _i]),
    "f2_filterbank_envelope_fused": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _d, _i, _vp, _vp, _i]),
    "f2_gather_windows": (_i, [_vp, _vp, _i, _i64, _vp, _i64, _i, _i, _i, _vp, _i]),
    "f2_cnn_create": (_i, [_vp, _P(_vp), _i, _i, _P(_vp)]),
    "f2_cnn_destroy": (_i, [_vp, _vp]),
    "f2_cnn_get_info": (_i, [_vp, _vp, C.c_char_p, _P(_d)]),
    "f2_cnn_forward": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i]),
    "f2_eval_utterance": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _i, _i, _d, _i, _i, _i, _vp, _vp, _vp, _P(_i64), _i]),
    "f2_eval_batch": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _d, _i, _i, _i, _vp, _vp, _i]),
}

_lib = None
_lock = threading.Lock()


class F2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libf2cnn_hip error {code}: {msg}")
        self.code = code


def library_path():
    return _build.LIB_PATH


def _share_hip_runtime_with_torch():
    """A process can hold one HIP runtime. PyTorch-ROCm ships its own libamdhip64.so.7 (same SONAME as the system
    one this library is linked against), and whichever loads first serves both: with the system runtime loaded first,
    a later `import torch` (bench.py's process group, `cnn train`) finds no usable device. So when torch is installed
    its runtime is mapped first, whatever the import order. F2CNN_HIP_RUNTIME=system skips this."""
    if os.environ.get("F2CNN_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load(build_if_missing=True):
    """dlopen the library (building it first if it is absent) and declare every signature."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            if not build_if_missing:
                raise FileNotFoundError(f"{path} is missing: run `python -m f2cnn_amd.build` (needs hipcc)")
            _build.build_library()
        _share_hip_runtime_with_torch()
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if os.environ.get("F2CNN_PROBE_OLD_LIB") == "1":   # tools/ab_old_new.sh: an older build beside the tree's
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a  # int device pointer


class _PinnedOwner:
    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr

    def __del__(self):
        try:
            if self.ctx.handle and self.ptr:
                self.ctx.lib.f2_host_free(self.ctx.handle, self.ptr)
        except Exception:
            pass


class Context:
    """One f2_ctx: a device, a stream and its scratch memory. Not thread-safe (one host thread per context)."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.f2_ctx_create(int(device), C.byref(h))
        if rc != F2_OK:
            raise F2Error(rc, self.lib.f2_last_error(None).decode())
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.f2_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != F2_OK:
            raise F2Error(rc, self.lib.f2_last_error(self.handle).decode())

    # ---- plumbing ----
    def synchronize(self):
        self.check(self.lib.f2_ctx_synchronize(self.handle))

    def set_stream(self, stream_ptr):
        self.check(self.lib.f2_ctx_set_stream(self.handle, stream_ptr))

    def set_option(self, key, value):
        """Per-context tuning switch (include/f2cnn_hip.h: f2_ctx_set_option); returns the previous value."""
        old = self.get_option(key)
        self.check(self.lib.f2_ctx_set_option(self.handle, key.encode(), float(value)))
        return old

    def get_option(self, key):
        v = _d()
        self.check(self.lib.f2_ctx_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def spectral_guard_values(self):
        """(rows, 4) float32 {row max, padding residual, low-passed row max, flagged} of the last fused call made with option
        spectral_guard_dump = 1, rows in (utterance, channel) batch order; NaN rows were not served by the spectral kernel."""
        avail = _i64()
        self.check(self.lib.f2_spectral_guard_read(self.handle, None, 0, C.byref(avail)))
        out = np.empty((avail.value, 4), np.float32)
        if avail.value:
            self.check(self.lib.f2_spectral_guard_read(self.handle, out.ctypes.data, avail.value, None))
        return out

    def options(self, **kv):
        """Context manager: set the given options, restore the previous values on exit."""
        ctx = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.old = {k: ctx.set_option(k, v) for k, v in kv.items()}
                return ctx

            def __exit__(self_inner, *exc):
                for k, v in self_inner.old.items():
                    ctx.set_option(k, v)
                return False

        return _Scope()

    def malloc(self, nbytes):
        p = _vp()
        self.check(self.lib.f2_dev_malloc(self.handle, int(nbytes), C.byref(p)))
        return p.value or 0

    def free(self, dptr):
        self.check(self.lib.f2_dev_free(self.handle, dptr))

    def host_alloc(self, nbytes):
        """Page-locked host buffer of nbytes: a ctypes byte array (buffer protocol) that frees itself when collected."""
        p = _vp()
        self.check(self.lib.f2_host_alloc(self.handle, int(nbytes), C.byref(p)))
        buf = (C.c_ubyte * int(nbytes)).from_address(p.value)
        buf._f2_owner = _PinnedOwner(self, p.value)
        return buf

    def memset(self, dptr, value, nbytes):
        self.check(self.lib.f2_dev_memset(self.handle, dptr, value, int(nbytes)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.f2_memcpy_h2d(self.handle, dptr, arr.ctypes.data, arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        self.check(self.lib.f2_memcpy_d2h(self.handle, arr.ctypes.data, dptr, arr.nbytes))

    def event(self):
        e = _vp()
        self.check(self.lib.f2_event_create(self.handle, C.byref(e)))
        return e

    def record(self, ev):
        self.check(self.lib.f2_event_record(self.handle, ev))

    def elapsed_ms(self, e0, e1):
        ms = C.c_float()
        self.check(self.lib.f2_event_elapsed_ms(self.handle, e0, e1, C.byref(ms)))
        return ms.value

    def event_done(self, ev):
        """True once the stream has passed the event (does not wait)."""
        done = C.c_int()
        self.check(self.lib.f2_event_query(self.handle, ev, C.byref(done)))
        return bool(done.value)

    def destroy_event(self, ev):
        self.check(self.lib.f2_event_destroy(self.handle, ev))

    def prof_enable(self, on=True):
        self.check(self.lib.f2_prof_enable(self.handle, int(on)))

    def prof_get(self):
        """{kernel name: (launches, total_ms)} since profiling was enabled/reset; waits for the stream."""
        out = {}
        for k in range(K_COUNT):
            n, ms = _i(), C.c_float()
            self.check(self.lib.f2_prof_get(self.handle, k, C.byref(n), C.byref(ms)))
            if n.value:
                out[self.lib.f2_prof_kernel_name(k).decode()] = (n.value, ms.value)
        return out

    def prof_reset(self):
        self.check(self.lib.f2_prof_reset(self.handle))

    # ---- ops (data pointers: numpy arrays with mem_space HOST, or int device pointers) ----
    def erb_filterbank_batch(self, wave, wave_dtype, offsets, coefs, B, Cn, gfb, mem_space):
        self.check(self.lib.f2_erb_filterbank_batch(self.handle, _ptr(wave), wave_dtype, _ptr(offsets), _ptr(coefs),
                                                    B, Cn, _ptr(gfb), mem_space))

    def envelope_batch(self, gfb, offsets, B, Cn, lpf, cutoff, precision, env, mem_space):
        self.check(self.lib.f2_envelope_batch(self.handle, _ptr(gfb), _ptr(offsets), B, Cn, int(bool(lpf)),
                                              float(cutoff), precision, _ptr(env), mem_space))

    def filterbank_envelope_fused(self, wave, wave_dtype, offsets, coefs, B, Cn, lpf, cutoff, precision, env, gfb,
                                  mem_space):
        self.check(self.lib.f2_filterbank_envelope_fused(self.handle, _ptr(wave), wave_dtype, _ptr(offsets),
                                                         _ptr(coefs), B, Cn, int(bool(lpf)), float(cutoff), precision,
                                                         _ptr(env), _ptr(gfb), mem_space))

    def gather_windows(self, env, Cn, N, centers, n_windows, radius, step, normalize, out, mem_space):
        self.check(self.lib.f2_gather_windows(self.handle, _ptr(env), Cn, N, _ptr(centers), n_windows, radius, step,
                                              int(bool(normalize)), _ptr(out), mem_space))

    def cnn_create(self, tensors, rows, channels):
        """tensors: 12 contiguous float32 arrays in Keras layouts (see include/f2cnn_hip.h). Returns a handle."""
        arrs = [np.ascontiguousarray(t, dtype=np.float32) for t in tensors]
        if len(arrs) != 12:
            raise ValueError("the network has 12 weight tensors")
        ptrs = (_vp * 12)(*[a.ctypes.data for a in arrs])
        h = _vp()
        self.check(self.lib.f2_cnn_create(self.handle, ptrs, int(rows), int(channels), C.byref(h)))
        return h

    def cnn_info(self, handle, key):
        """f2_cnn_get_info: 'ws_ok', 'ws_dense_ok', 'ws_check_diff', 'ws_dense_check_diff', 'flat'"""
        v = _d()
        self.check(self.lib.f2_cnn_get_info(self.handle, handle, key.encode(), C.byref(v)))
        return v.value

    def cnn_destroy(self, handle):
        self.check(self.lib.f2_cnn_destroy(self.handle, handle))

    def cnn_forward(self, handle, x, n, scores, labels, mem_space):
        self.check(self.lib.f2_cnn_forward(self.handle, handle, _ptr(x), int(n), _ptr(scores), _ptr(labels), mem_space))

    def eval_utterance(self, handle, wave, wave_dtype, N, coefs, Cn, lpf, cutoff, precision, radius, step, env, scores,
                       labels, mem_space):
        nb = _i64()
        self.check(self.lib.f2_eval_utterance(self.handle, handle, _ptr(wave), wave_dtype, int(N), _ptr(coefs), Cn,
                                              int(bool(lpf)), float(cutoff), precision, radius, step, _ptr(env),
                                              _ptr(scores), _ptr(labels), C.byref(nb), mem_space))
        return nb.value


    def eval_batch(self, handle, wave, wave_dtype, offsets, coefs, B, Cn, lpf, cutoff, precision, radius, step, scores,
                   labels, mem_space):
        """Ragged batch through filterbank, envelope, every-sample windows and the CNN; see f2_eval_batch."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.check(self.lib.f2_eval_batch(self.handle, handle, _ptr(wave), wave_dtype, _ptr(offsets), _ptr(coefs), int(B),
                                          Cn, int(bool(lpf)), float(cutoff), precision, radius, step, _ptr(scores),
                                          _ptr(labels), mem_space))

_default_ctx = {}
_extra_ctx = {}


def default_context(device=None):
    """Process-wide context per device (default: $F2CNN_DEVICE, else torchrun's $LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("F2CNN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    ctx = _default_ctx.get(device)
    if ctx is None or ctx.handle is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


def pipeline_contexts(n=2, device=None):
    """n contexts (= n streams with their own staging memory) on one device, the first being the default context:
    the file drivers alternate batches between them so that one batch's copies run beside the next one's kernels."""
    first = default_context(device)
    out = [first]
    for i in range(1, n):
        key = (first.device, i)
        ctx = _extra_ctx.get(key)
        if ctx is None or ctx.handle is None:
            ctx = _extra_ctx[key] = Context(first.device)
        out.append(ctx)
    return out
