// Register-level FFT building blocks shared by the LDS-resident envelope kernel (f2_envelope.hip) and the
// global-memory path for long rows (f2_envelope_large.hip).
#pragma once
#include "f2_internal.h"

namespace f2fft {

using EnvParams = ::f2_env_params;   // kernel arguments shared by the envelope kernels (f2_internal.h)


template <typename F>
struct cpx {
    F re, im;
};
template <typename F>
__device__ __forceinline__ cpx<F> operator+(cpx<F> a, cpx<F> b) { return {a.re + b.re, a.im + b.im}; }
template <typename F>
__device__ __forceinline__ cpx<F> operator-(cpx<F> a, cpx<F> b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ float fsqrt(float v) { return __builtin_amdgcn_sqrtf(v); }   // v_sqrt_f32, 1 ulp
// float64 magnitude: v_rsq_f64 (~2^-26), one Goldschmidt step (2^-52) and the residual correction - within an ulp or two of
// sqrt for every normal argument (the envelope is held to 1e-10; the correctly rounded library sqrt adds a second step, argument
// scaling and class tests: ~25 instructions where this has 8, a sixth of the float64 arithmetic of the float64 envelope kernel)
__device__ __forceinline__ double fsqrt(double s) {
    const double r = __builtin_amdgcn_rsq(s);
    double g = s * r, h = 0.5 * r;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, s);
    g = fma(d, h, g);
    return s > 0.0 ? g : 0.0;      // (s = 0: rsq = inf, 0 x inf)
}
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

// 16-byte store of two consecutive float64 results, non-temporal: the envelopes are written once and not read again by
// this kernel, and keeping them out of the way of the cache's other lines is worth 4-5 % of K2 (6.14 -> 5.85 ms together
// with non-temporal loads of the hand-off; -DF2_TEMPORAL restores plain accesses).
typedef double f2_d2 __attribute__((ext_vector_type(2)));
typedef float f2_f2 __attribute__((ext_vector_type(2)));
// A row of a (C, n) matrix starts on an element boundary, not on a pair boundary (odd n: every other row). Global
// memory takes dword-aligned wide accesses, so pairs are always moved as one 16-byte (float64) or 8-byte (float32)
// access through these under-aligned types, whatever the parity of the row.
typedef f2_d2 f2_d2u __attribute__((aligned(8)));
typedef f2_f2 f2_f2u __attribute__((aligned(4)));
__device__ __forceinline__ void store_pair(double* p, double a, double b) {
    f2_d2u v = {a, b};
#ifdef F2_TEMPORAL
    *reinterpret_cast<f2_d2u*>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<f2_d2u*>(p));
#endif
}
__device__ __forceinline__ f2_d2 load_pair_f64(const double* p) {
#ifdef F2_TEMPORAL
    return *reinterpret_cast<const f2_d2u*>(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const f2_d2u*>(p));
#endif
}
__device__ __forceinline__ f2_f2 load_pair_f32(const float* p) {
#ifdef F2_TEMPORAL
    return *reinterpret_cast<const f2_f2u*>(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const f2_f2u*>(p));
#endif
}
// Samples i0 and i0 + 1 (i0 even, any value >= 0) of a row of n >= 2 samples, zero beyond its end: one wide load from
// an address clamped into the row, no divergent loads. ODD = false is the cheaper form for rows of even length (a
// pair is inside the row or outside it); with ODD the last sample of an odd-length row arrives as the second half of
// the pair (n-2, n-1). Callers branch on the parity of n once, outside their unrolled loops.
template <bool ODD, typename F>
__device__ __forceinline__ void row_pair(const float* __restrict__ x, int n, int i0, F& a, F& b) {
    const int p = min(i0, n - 2);
    const f2_f2 t = load_pair_f32(x + p);
    a = i0 < n ? (F)((!ODD || p == i0) ? t.x : t.y) : F(0);
    b = (ODD ? i0 + 1 < n : i0 < n) ? (F)t.y : F(0);
}
template <bool ODD, typename F>
__device__ __forceinline__ void row_pair(const double* __restrict__ x, int n, int i0, F& a, F& b) {
    const int p = min(i0, n - 2);
    const f2_d2 t = load_pair_f64(x + p);
    a = i0 < n ? (F)((!ODD || p == i0) ? t.x : t.y) : F(0);
    b = (ODD ? i0 + 1 < n : i0 < n) ? (F)t.y : F(0);
}
// envelope samples i0, i0 + 1 of a row of n samples
__device__ __forceinline__ void store_row_pair(double* __restrict__ y, int n, int i0, double a, double b) {
    if (i0 + 1 < n) store_pair(y + i0, a, b);
    else if (i0 < n) y[i0] = a;
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


// ---- first-order low-pass in the register layout of the FFT, followed by the float64 stores ----
// Thread t holds the envelope pairs (2m, 2m+1), m = t + NT*jj, jj < NBLK: one pair in each of NBLK blocks of 2*NT
// consecutive samples (er[jj] = env[2m], ei[jj] = env[2m+1]).
//   y[n] = q y[n-1] + u[n],  u[n] = b0 (e[n] + e[n-1]),  q = -a1,  zero initial state
// pair:  z0 = u0, z1 = u1 + q u0                      (zero state at the pair start)
// block: inclusive scan of z1 over the NT pairs with multiplier q^2 (DPP wave scan in F, then the wave totals and
//        everything above them in float64)
// row:   the NBLK block totals are chained sequentially (q^(2 NT) per block)
// and the outputs leave as coalesced 16-byte stores straight from the registers. All threads of the workgroup must
// call it, after a barrier that makes `smem` (lowpass_lds_bytes bytes) free.
template <typename F, int NT, int NBLK>
constexpr size_t lowpass_lds_bytes() {
    return sizeof(F) * NBLK * NT + sizeof(double) * (2 * NBLK * (NT / 64) + NBLK);
}

// y_in / e_in: filter state entering the first sample (y[-1], e[-1]); the return value is y of the last of the
// 2*NT*NBLK samples (rows longer than that are filtered segment by segment, f2_envelope_split.hip).
template <typename F, int NT, int NBLK>
__device__ __forceinline__ double lowpass_pairs_store(const F (&er)[NBLK], const F (&ei)[NBLK], double a1, double b0,
                                                      unsigned char* smem, double* __restrict__ y, int n, int tid,
                                                      double y_in = 0.0, F e_in = F(0)) {
    constexpr int NW = NT / 64;
    static_assert(NBLK <= NT, "one thread per block chains the wave totals");
    F* e1s = reinterpret_cast<F*>(smem);                           // [NBLK][NT] odd samples, for e[n-1]
    double* wtot = reinterpret_cast<double*>(smem + sizeof(F) * NBLK * NT);   // [NBLK][NW] wave totals
    double* cwl = wtot + NBLK * NW;                                // [NBLK][NW] carry into each wave
    double* btot = cwl + NBLK * NW;                                // [NBLK] block totals
    const int lane = tid & 63, wv = tid >> 6;
    const double q = -a1;
    const F qf = (F)q, b0f = (F)b0;
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) e1s[jj * NT + tid] = ei[jj];
    // powers of q^2 for the wave scan, q^(2(lane+1)), q^128, q^(2 tid), q^(2 NT)
    double g[6];
    g[0] = q * q;
#pragma unroll
    for (int d = 1; d < 6; ++d) g[d] = g[d - 1] * g[d - 1];
    const double gw = g[5] * g[5];                                 // q^128: one wave of pairs
    double gl = 1.0;                                               // q^(2 (lane+1))
    {
        double gp = g[0];
        for (int bits = lane + 1; bits; bits >>= 1) {
            if (bits & 1) gl *= gp;
            gp *= gp;
        }
    }
    double gt = 1.0, gblk = 1.0;                                   // q^(2 tid), q^(2 NT)
    {
        double gp = g[0];
        for (int bits = tid; bits; bits >>= 1) {
            if (bits & 1) gt *= gp;
            gp *= gp;
        }
        gp = g[0];
        for (int bits = NT; bits; bits >>= 1) {
            if (bits & 1) gblk *= gp;
            gp *= gp;
        }
    }
    __syncthreads();
    F u0[NBLK], u1[NBLK], sc[NBLK];
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) {
        const F eprev = tid > 0 ? e1s[jj * NT + tid - 1] : (jj > 0 ? e1s[(jj - 1) * NT + NT - 1] : e_in);
        u0[jj] = b0f * (er[jj] + eprev);
        u1[jj] = b0f * (ei[jj] + er[jj]);
        sc[jj] = u1[jj] + qf * u0[jj];
    }
    // inclusive weighted scan over the 64 pairs of the wave (all NBLK blocks interleaved) on DPP lane moves:
    // row_shr 1,2,4,8 inside each row of 16 lanes (lanes without a source read 0), then row_bcast:15 brings the
    // previous row's total into rows 1 and 3, and row_bcast:31 lane 31's total into rows 2 and 3
    {
        const F g1 = (F)g[0], g2 = (F)g[1], g4 = (F)g[2], g8 = (F)g[3];
        F m16 = F(1), m32 = F(1);                                  // (q^2)^((lane&15)+1), (q^2)^((lane&31)+1)
        {
            double a = 1.0, b = 1.0, gp = g[0];
            for (int bit = 0; bit < 5; ++bit) {
                if ((((lane & 15) + 1) >> bit) & 1) a *= gp;
                if ((((lane & 31) + 1) >> bit) & 1) b *= gp;
                gp *= gp;
            }
            b = (((lane & 31) + 1) >> 5) & 1 ? b * gp : b;
            m16 = (F)a;
            m32 = (F)b;
        }
#pragma unroll
        for (int jj = 0; jj < NBLK; ++jj) {
            sc[jj] += g1 * dpp_mov<0x111, 0xF>(sc[jj]);
            sc[jj] += g2 * dpp_mov<0x112, 0xF>(sc[jj]);
            sc[jj] += g4 * dpp_mov<0x114, 0xF>(sc[jj]);
            sc[jj] += g8 * dpp_mov<0x118, 0xF>(sc[jj]);
            sc[jj] += m16 * dpp_mov<0x142, 0xA>(sc[jj]);
            sc[jj] += m32 * dpp_mov<0x143, 0xC>(sc[jj]);
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int jj = 0; jj < NBLK; ++jj) wtot[jj * NW + wv] = (double)sc[jj];
    }
    __syncthreads();
    // one thread per block chains the NW wave totals: cwl[jj][w] = zero-state value of block jj at the end of
    // wave w-1, btot[jj] = at the end of the block
    if (tid < NBLK) {
        double c = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
            const double t = wtot[tid * NW + w2];
            cwl[tid * NW + w2] = c;
            c = fma(gw, c, t);
        }
        btot[tid] = c;
    }
    __syncthreads();
    double ycarry = y_in;                                          // true y at the end of the previous block
    const F glf = (F)gl, gtf = (F)gt;
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) {
        // the carries into the wave (cw) and into the block (ycarry) were accumulated in float64; from here on the
        // values are combined in F: they are sums of non-negative terms of the size of the output itself
        const F cw = (F)cwl[jj * NW + wv];
        const F sin_ = glf * cw + sc[jj];                          // zero-state value at the end of this pair
        const F up = dpp_mov<0x138, 0xF>(sin_);                    // wave_shr:1
        const F sprev = lane > 0 ? up : cw;                        // ... at the end of the previous pair
        const F y0 = qf * (gtf * (F)ycarry + sprev) + u0[jj];
        const F y1 = qf * y0 + u1[jj];
        const int i0 = 2 * (tid + NT * jj);
        // (a block that lies inside the row - a wave-uniform test - stores without a test per lane)
        if (2 * NT * (jj + 1) <= n) store_pair(y + i0, (double)y0, (double)y1);
        else if (2 * NT * jj < n) store_row_pair(y, n, i0, (double)y0, (double)y1);
        ycarry = fma(gblk, ycarry, btot[jj]);
    }
    return ycarry;
}

}  // namespace f2fft
