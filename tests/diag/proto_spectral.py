"""Numerical prototype of the spectral filterbank + envelope path (kernel KS).

The envelope of channel c needs  a = analytic_M(y_c zero-padded to M),  y_c = erb_filterbank(x)[c].  Instead of running
the recurrence over every sample and transforming the result, the spectrum of the zero-padded output is formed
directly:

    Y_c(k) = H_c(z_k) X(k) + z_k^-N u^4 Q(z_k) / gain,     z_k = exp(2 pi i k / M),  u = 1 / D(z_k)

X = DFT_M(x zero-padded), H_c = prod_m N_m / D^4 / gain the cascade's frequency response, and the second term the
spectrum of (minus) the M-periodised ringing after sample N: the recurrence D y_m = N_m y_(m-1) holds for n < N, so
for the truncated sequences  D Y_m = N_m Y_(m-1) + z^-N R_m(z)  with R_m of degree <= 1 built from the last two
outputs of sections m and m-1.  Q = R_4 D^3 + N_4 R_3 D^2 + N_4 N_3 R_2 D + N_4 N_3 N_2 R_1 is kept as its D-adic
digits  Q = sum_j r_j D^j (deg r_j <= 1)  so that  u^4 Q = r_3 u + r_2 u^2 + r_1 u^3 + r_0 u^4  is a Horner chain in u
without cancellation.  The last outputs of the sections only depend on the last L samples of x (the poles decay).

The analytic signal of the one-sided spectrum splits into the even and the odd output samples: two H = M/2 point
complex transforms, A_e(k) = A(k), A_o(k) = A(k) exp(i pi k / H)  (k = 0 takes the Nyquist term +-A(H)).

Run:  python tests/diag/proto_spectral.py   (CPU only; float32 emulation of the device arithmetic)
"""
import sys
import numpy as np

sys.path.insert(0, '.')
from oracle import f2cnn_oracle as orc   # noqa: E402  (prototype = test infrastructure)

F32 = np.float32
C64 = np.complex64


def section_polys(row):
    A0, A11, A12, A13, A14, A2, B0, B1, B2, gain = [float(v) for v in row]
    Ns = [np.array([A0, a1, A2]) / B0 for a1 in (A11, A12, A13, A14)]
    D = np.array([1.0, B1 / B0, B2 / B0])
    return Ns, D, gain


def run_sections_tail(x, row, L):
    """Last two outputs of the input and of each section, running only the last L samples from zero state."""
    from scipy.signal import lfilter
    Ns, D, gain = section_polys(row)
    seg = np.asarray(x[-L:], dtype=np.float64) if L < len(x) else np.asarray(x, dtype=np.float64)
    outs = [seg]
    y = seg
    for Nm in Ns:
        y = lfilter(Nm, D, y)
        outs.append(y)
    last = []
    for o in outs:
        o2 = np.concatenate([np.zeros(2), o])[-2:]
        last.append((o2[1], o2[0]))       # (y[N-1], y[N-2])
    return last


def polymul(a, b):
    return np.convolve(a, b)


def polydivD(p, D):
    """p(w) = a(w) D(w) + r(w), deg r <= 1 (polynomials in w = z^-1, lowest power first)."""
    p = np.array(p, dtype=np.float64)
    if len(p) <= 2:
        return np.zeros(1), np.concatenate([p, np.zeros(2 - len(p))])
    a = np.zeros(len(p) - 2)
    for i in range(len(p) - 1, 1, -1):
        q = p[i] / D[2]
        a[i - 2] = q
        p[i - 2:i + 1] -= q * D
    return a, p[:2]


def dadic_digits(row, last):
    """r_0..r_3 (each (rho0, rho1)) of Q = R4 D^3 + N4 R3 D^2 + N4 N3 R2 D + N4 N3 N2 R1."""
    Ns, D, gain = section_polys(row)
    R = []
    for m in range(4):
        (ym1, ym2), (xm1, xm2) = last[m + 1], last[m]
        Nm = Ns[m]
        # R_m(i) = [D*y_m - N_m*y_(m-1)](N+i), sequences zero from N on
        r0 = D[1] * ym1 + D[2] * ym2 - Nm[1] * xm1 - Nm[2] * xm2
        r1 = D[2] * ym1 - Nm[2] * xm1
        R.append(np.array([r0, r1]))
    digits = [np.zeros(2) for _ in range(4)]
    terms = [(0, polymul(polymul(polymul(Ns[3], Ns[2]), Ns[1]), R[0])),
             (1, polymul(polymul(Ns[3], Ns[2]), R[1])),
             (2, polymul(Ns[3], R[2])),
             (3, R[3])]
    for shift, p in terms:
        j = shift
        while True:
            a, r = polydivD(p, D)
            digits[j] += r
            if not np.any(a):
                break
            p = a
            j += 1
            assert j < 4 or not np.any(np.abs(a) > 0) or True
            if j >= 4:
                # degree bookkeeping: total degree <= 7, so this must be zero
                assert np.allclose(a, 0, atol=0), a
                break
    return digits


def tail_len(row, fs_tol=1e-10):
    """samples after which the cascade's impulse response envelope has decayed below fs_tol of its peak"""
    _, D, _ = section_polys(row)
    r = np.sqrt(D[2])
    n = np.arange(1, 200000)
    env = n ** 3.0 * r ** n
    pk = env.max()
    idx = np.nonzero(env > fs_tol * pk)[0]
    return int(idx[-1]) + 64


def tables(coefs, M):
    H = M // 2
    k = np.arange(H + 1)
    w = np.exp(-2j * np.pi * k / M)       # z^-1
    Ht = np.zeros((len(coefs), H + 1), dtype=np.complex128)
    ut = np.zeros_like(Ht)
    for c, row in enumerate(coefs):
        Ns, D, gain = section_polys(row)
        Dz = D[0] + D[1] * w + D[2] * w * w
        u = 1.0 / Dz
        h = np.ones_like(u)
        for Nm in Ns:
            h = h * (Nm[0] + Nm[1] * w + Nm[2] * w * w) * u
        Ht[c] = h / gain
        ut[c] = u
    return w, Ht, ut


def spectral_envelope(x, coefs, f32=True, tail_tol=1e-10, fft32=True):
    N = len(x)
    M = orc.padded_length(N)
    H = M // 2
    xp = np.zeros(M)
    xp[:N] = x
    X = np.fft.rfft(xp)                    # float64 forward transform of the utterance (once per utterance)
    w, Ht, ut = tables(coefs, M)
    k = np.arange(H + 1)
    zeta = np.exp(-2j * np.pi * ((k * N) % M) / M)
    cplx = C64 if f32 else np.complex128
    real = F32 if f32 else np.float64
    Xc, wc, zc = X.astype(cplx), w.astype(cplx), zeta.astype(cplx)
    env = np.zeros((len(coefs), N))
    for c, row in enumerate(coefs):
        Ns, D, gain = section_polys(row)
        L = tail_len(row, tail_tol)
        last = run_sections_tail(x, row, L)
        dg = dadic_digits(row, last)
        rho = [(real(d[0] / gain), real(d[1] / gain)) for d in dg]
        Hc, uc = Ht[c].astype(cplx), ut[c].astype(cplx)

        def rj(j):
            return (rho[j][0] + rho[j][1] * wc).astype(cplx)
        t = rj(0)
        t = (rj(1) + uc * t).astype(cplx)
        t = (rj(2) + uc * t).astype(cplx)
        t = (rj(3) + uc * t).astype(cplx)
        Y = (Xc * Hc + zc * (uc * t).astype(cplx)).astype(cplx)
        A = 2 * Y
        A[0] = Y[0]
        A[H] = Y[H]
        Ae = A[:H].copy()
        Ao = (A[:H] * np.conj(wc[:H])).astype(cplx)   # exp(+i pi k / H) = conj(w)
        Ae[0] = A[0] + A[H]
        Ao[0] = A[0] - A[H]
        if f32 and fft32:
            import scipy.fft as sfft
            ae = sfft.ifft(Ae.astype(C64))
            ao = sfft.ifft(Ao.astype(C64))
            assert ae.dtype == C64
        else:
            ae = np.fft.ifft(Ae.astype(np.complex128))
            ao = np.fft.ifft(Ao.astype(np.complex128))
        a = np.empty(M, dtype=np.complex128)
        a[0::2] = ae / 2
        a[1::2] = ao / 2
        env[c] = np.abs(a[:N])
    return env


def relerr(a, b):
    return np.max(np.abs(a - b), axis=1) / np.max(np.abs(b), axis=1)


def main():
    fs = 16000
    rng = np.random.default_rng(2027)
    C = 128
    coefs = orc.make_erb_filters(fs, orc.centre_freqs(fs, C, 100))
    chans = list(range(0, C, 9)) + [C - 3, C - 2, C - 1]
    sub = coefs[chans]
    cases = {}
    cases['noise16000'] = np.clip(np.round(rng.standard_normal(16000) * 3000), -32768, 32767)
    cases['noise15999'] = np.clip(np.round(rng.standard_normal(15999) * 3000), -32768, 32767)
    cases['noise9000'] = np.clip(np.round(rng.standard_normal(9000) * 3000), -32768, 32767)
    cases['noise16384'] = np.clip(np.round(rng.standard_normal(16384) * 3000), -32768, 32767)
    t = np.arange(16000) / fs
    cases['sine1k'] = np.round(10000 * np.sin(2 * np.pi * 1000 * t))
    cases['lowsine+hiss'] = np.round(20000 * np.sin(2 * np.pi * 120 * t) + 3 * rng.standard_normal(16000))
    imp = np.zeros(16000)
    imp[0] = 32767
    cases['impulse0'] = imp
    imp2 = np.zeros(16000)
    imp2[15990] = 32767
    cases['impulse_end'] = imp2
    cases['dc'] = np.full(16000, 12000.0)
    # speech-like tilt: noise through a steep low-pass
    from scipy.signal import lfilter
    tilt = lfilter([1.0], [1.0, -0.98], rng.standard_normal(16000))
    cases['tilted'] = np.round(tilt / np.abs(tilt).max() * 30000)
    for name, x in cases.items():
        gfb = orc.erb_filterbank(x, sub)
        ref = np.abs(np.array([orc.padded_hilbert(r) for r in gfb]))
        for label, kw in (('f64', dict(f32=False)), ('f32', dict(f32=True))):
            env = spectral_envelope(x, sub, **kw)
            e = relerr(env, ref)
            print(f'{name:14s} {label}: max rel err {e.max():.3e} (chan {chans[int(np.argmax(e))]}), median {np.median(e):.2e}')
        # with low-pass 50 Hz
        refl = np.array([orc.low_pass_filter(r, 50) for r in ref])
        envl = np.array([orc.low_pass_filter(r, 50) for r in env])
        print(f'{"":14s} f32 + LPF50: {relerr(envl, refl).max():.3e}')


if __name__ == '__main__':
    main()
