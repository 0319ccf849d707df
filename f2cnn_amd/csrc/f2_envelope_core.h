// Register-level FFT building blocks shared by the LDS-resident envelope kernel (f2_envelope.hip) and the
// global-memory path for long rows (f2_envelope_large.hip).
#pragma once
#include "f2_internal.h"

namespace f2fft {

template <typename F>
struct cpx {
    F re, im;
};
template <typename F>
__device__ __forceinline__ cpx<F> operator+(cpx<F> a, cpx<F> b) { return {a.re + b.re, a.im + b.im}; }
template <typename F>
__device__ __forceinline__ cpx<F> operator-(cpx<F> a, cpx<F> b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ float fsqrt(float v) { return __builtin_amdgcn_sqrtf(v); }   // v_sqrt_f32, 1 ulp
__device__ __forceinline__ double fsqrt(double v) { return sqrt(v); }
template <typename F>
__device__ __forceinline__ cpx<F> cmul(cpx<F> a, cpx<F> w) {
    return {a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re};
}

// cos/sin(2 pi j / 32), j = 0..15
__device__ constexpr double kCos32[16] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                          0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                                          0.19509032201612826785, 0.0, -0.19509032201612826785, -0.38268343236508977173,
                                          -0.55557023301960222474, -0.70710678118654752440, -0.83146961230254523708,
                                          -0.92387953251128675613, -0.98078528040323044913};
__device__ constexpr double kSin32[16] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                                          0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                                          0.98078528040323044913, 1.0, 0.98078528040323044913, 0.92387953251128675613,
                                          0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                                          0.38268343236508977173, 0.19509032201612826785};

// a * exp(-2 pi i K / R)
template <int R, int K, typename F>
__device__ __forceinline__ cpx<F> mulw(cpx<F> a) {
    if constexpr (K == 0) {
        return a;
    } else if constexpr (4 * K == R) {
        return {a.im, -a.re};
    } else if constexpr (8 * K == R) {
        const F h = F(0.70710678118654752440);
        return {(a.re + a.im) * h, (a.im - a.re) * h};
    } else if constexpr (8 * K == 3 * R) {
        const F h = F(0.70710678118654752440);
        return {(a.im - a.re) * h, -(a.re + a.im) * h};
    } else {
        const F c = F(kCos32[K * 32 / R]), s = F(kSin32[K * 32 / R]);
        return {a.re * c + a.im * s, a.im * c - a.re * s};
    }
}

template <int R, int J, typename F>
__device__ __forceinline__ void dif_stage(cpx<F>* v) {
    if constexpr (J < R / 2) {
        const cpx<F> a = v[J], b = v[J + R / 2];
        v[J] = a + b;
        v[J + R / 2] = mulw<R, J>(a - b);
        dif_stage<R, J + 1>(v);
    }
}

// forward DFT of R points held in registers, decimation in frequency, in place (no register shuffling):
// natural order in, BIT-REVERSED order out: X[k] is left in v[brev<R>(k)]
template <int R, typename F>
__device__ __forceinline__ void dft(cpx<F>* v) {
    if constexpr (R >= 2) {
        dif_stage<R, 0>(v);
        dft<R / 2>(v);
        dft<R / 2>(v + R / 2);
    }
}

template <int R>
constexpr int brev(int k) {
    int r = 0;
    for (int b = 1; b < R; b <<= 1) {
        r = (r << 1) | (k & 1);
        k >>= 1;
    }
    return r;
}

__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, d);
    hi = __shfl_up(hi, d);
    return __hiloint2double(hi, lo);
}

// DPP lane move (gfx9 encodings: 0x110+n row_shr:n, 0x142 row_bcast:15, 0x143 row_bcast:31); lanes without a source
// lane, or in rows masked out by ROW_MASK, receive 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo);
}

}  // namespace f2fft
