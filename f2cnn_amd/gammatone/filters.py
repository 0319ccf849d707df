"""Drop-in for the reference's ``gammatone/filters.py`` on the hot path.

``centre_freqs`` / ``make_erb_filters`` (reference gammatone/filters.py:74-86, :89-192) run once per
job on C <= a few hundred channels and stay on the host in float64; ``erb_filterbank`` (:195-239) is
the HIP kernel K1 reached through ``f2_erb_filterbank_batch``.
"""
import numpy as np

from .. import _lib

# Glasberg & Moore ERB scale constants used by the reference (gammatone/filters.py:36-38)
_EAR_Q = 9.26449
_MIN_BW = 24.7
_BW_CORRECTION = 1.019


def erb_point(low_freq, high_freq, fraction):
    """Point on the ERB scale: ``high_freq`` at fraction 0, ``low_freq`` at fraction 1 (filters.py:21-52)."""
    k = _EAR_Q * _MIN_BW
    span = np.log(low_freq + k) - np.log(high_freq + k)
    return (high_freq + k) * np.exp(np.asarray(fraction) * span) - k


def erb_space(low_freq=100, high_freq=44100 / 4, num=100):
    """``num`` frequencies from just below ``high_freq`` down to ``low_freq`` (filters.py:55-71)."""
    return erb_point(low_freq, high_freq, np.arange(1, num + 1) / num)


def centre_freqs(fs, num_freqs, cutoff):
    """Centre frequencies for :func:`make_erb_filters`, descending, last == ``cutoff`` (filters.py:74-86)."""
    return erb_space(cutoff, fs / 2, num_freqs)


def make_erb_filters(fs, centre_freqs, width=1.0):
    """(C,10) float64 rows ``[A0, A11, A12, A13, A14, A2, B0, B1, B2, gain]`` (filters.py:131-192).

    Each channel is four second-order sections with common poles ``r*exp(+-j*theta)``,
    ``r = exp(-B*T)``, ``theta = 2*pi*cf*T``, and one real zero each at
    ``r*(cos(theta) +- sqrt(3 +- 2**1.5)*sin(theta))``; ``gain`` is the cascade's magnitude at cf.
    """
    cf = np.atleast_1d(np.asarray(centre_freqs, dtype=np.float64))
    T = 1.0 / fs
    bw = _BW_CORRECTION * 2 * np.pi * width * (cf / _EAR_Q + _MIN_BW)
    theta = 2 * np.pi * cf * T
    r = np.exp(-bw * T)
    cs, sn = np.cos(theta), np.sin(theta)
    s_hi, s_lo = np.sqrt(3 + 2 ** 1.5), np.sqrt(3 - 2 ** 1.5)
    zeros = np.stack([cs + s_hi * sn, cs - s_hi * sn, cs + s_lo * sn, cs - s_lo * sn])  # (4,C), per unit r
    coefs = np.empty((cf.shape[0], 10))
    coefs[:, 0] = T
    coefs[:, 1:5] = (-T * r * zeros).T
    coefs[:, 5] = 0.0
    coefs[:, 6] = 1.0
    coefs[:, 7] = -2 * cs / np.exp(bw * T)
    coefs[:, 8] = np.exp(-2 * bw * T)
    e2 = np.exp(2j * theta)
    pole = np.exp(1j * theta - bw * T)
    numer = np.prod(e2[None, :] - pole[None, :] * zeros, axis=0)
    denom = -1 / np.exp(bw * T) + 1 + e2 * (1 - np.exp(bw * T))
    coefs[:, 9] = np.abs(numer * (T * np.exp(bw * T) / denom) ** 4)
    return coefs


def _wave_args(wave):
    wave = np.asarray(wave)
    if wave.ndim != 1:
        raise ValueError("wave must be a one dimensional sequence")
    if wave.dtype == np.int16:
        return np.ascontiguousarray(wave), _lib.WAVE_I16
    # anything else (the evalnoise path hands over float64) is filtered as float64, like lfilter does
    return np.ascontiguousarray(wave, dtype=np.float64), _lib.WAVE_F64


def erb_filterbank(wave, coefs, ctx=None):
    """Gammatone filterbank of one waveform: (C,N) float64, one channel per row (filters.py:195-239)."""
    ctx = ctx or _lib.default_context()
    wave, dt = _wave_args(wave)
    coefs = np.ascontiguousarray(coefs, dtype=np.float64)
    if coefs.ndim != 2 or coefs.shape[1] != 10:
        raise ValueError("coefs must have shape (C,10) as returned by make_erb_filters")
    Cn, N = coefs.shape[0], wave.shape[0]
    out = np.empty((Cn, N), dtype=np.float64)
    offsets = np.array([0, N], dtype=np.int64)
    ctx.erb_filterbank_batch(wave, dt, offsets, coefs, 1, Cn, out, _lib.MEM_HOST)
    return out


def erb_filterbank_batch(waves, coefs, ctx=None, alloc=np.empty, wait=True):
    """List of waveforms (same dtype, any lengths) -> list of (C,N_b) matrices in one launch. `alloc(count, dtype=...)`
    provides the output buffer (the file drivers pass a pool of page-locked buffers). wait=False queues the copies
    and the kernel on the context's stream (F2_MEM_HOST_ASYNC) and returns a function that waits for them and
    returns the list."""
    ctx = ctx or _lib.default_context()
    coefs = np.ascontiguousarray(coefs, dtype=np.float64)
    if not len(waves):
        return [] if wait else (lambda: [])
    args = [_wave_args(w) for w in waves]
    dt = args[0][1] if all(a[1] == args[0][1] for a in args) else _lib.WAVE_F64
    dtype = np.int16 if dt == _lib.WAVE_I16 else np.float64
    offsets = np.zeros(len(args) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([a[0].shape[0] for a in args])
    flat = np.concatenate([a[0].astype(dtype, copy=False) for a in args]) if offsets[-1] else np.zeros(0, dtype)
    Cn = coefs.shape[0]
    out = alloc(Cn * int(offsets[-1]), dtype=np.float64)
    ctx.erb_filterbank_batch(flat, dt, offsets, coefs, len(args), Cn, out, _lib.MEM_HOST if wait else _lib.MEM_HOST_ASYNC)

    def finish(keep=(flat, offsets, coefs)):      # the host arrays of the queued copies live as long as this closure
        ctx.synchronize()
        return [out[Cn * offsets[b]:Cn * offsets[b + 1]].reshape(Cn, -1) for b in range(len(args))]
    return finish() if wait else finish
