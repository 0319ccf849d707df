// KS -- filterbank + Hilbert envelope (+ low-pass) of a row WITHOUT the time-domain filterbank output: the spectrum of
// the zero-padded channel signal is formed directly from the spectrum of the utterance, so the (C, n) filterbank matrix
// never exists - neither in HBM nor on chip. Same results as K1 -> K2 (reference: gammatone/filters.py:195-239
// erb_filterbank followed by scripts/processing/EnvelopeExtraction.py:20-67) within the float32-FFT tolerance.
//
// Mathematics (prototype with the derivation checked against the oracle: tests/diag/proto_spectral.py).  Row = channel c of
// an utterance of n samples, M = 2^ceil(log2 n), H = M/2, z_k = exp(2 pi i k / M), w = 1/z_k.  The cascade of the four
// second-order sections  D(w) y_m = N_m(w) y_(m-1),  D = 1 + a1 w + a2 w^2,  N_m = 1 + c_m w  (make_erb_filters: shared
// poles, no w^2 term in the numerators; overall factor s = (A0/B0)^4 / gain) holds for the samples n' < n; for the
// sequences cut at n it leaves a residue at n, n+1:   D Y_m = N_m Y_(m-1) + w^n R_m(w),  deg R_m <= 1, a linear function
// of the section's two state words at sample n.  Hence, with u = 1/D(w_k) and X = DFT_M(x zero-padded),
//
//     Y(k) = s [ u^4 N_1 N_2 N_3 N_4 X(k)  +  w^n u^4 Q(w) ],   Q = R_4 D^3 + N_4 R_3 D^2 + N_4 N_3 R_2 D + N_4 N_3 N_2 R_1
//
// is the DFT of the zero-padded filterbank row: the first term is the M-periodic steady response, the second removes
// the ringing after sample n (which a circular product alone would wrap around). Q is evaluated through its D-adic
// digits  Q = r_0 + r_1 D + r_2 D^2 + r_3 D^3  (deg r_j <= 1):  u^4 Q = u (r_3 + u (r_2 + u (r_1 + u r_0))), a Horner
// chain without cancellation (the monomial form of Q loses 1/|D|^3 ~ 1e9 near the resonance of the low channels).
// The state words at sample n only depend on the last L samples of the utterance (poles of radius r: n^3 r^n < 1e-8
// of its peak beyond L; L = 130 ... 2100 samples for the 128-channel bank): k_tail_state runs the float64 recurrences
// of K1 over those samples only and leaves the eight digit coefficients of every row. (The kernel is bound by float64
// throughput, not by the length of its serial runs: splitting every run into four chained segments made it slower.)
//
// The analytic signal a = IDFT_M of the one-sided spectrum A (A(k) = 2 Y(k), 0 < k < H; Y(0); Y(H)) is computed for the
// even and the odd samples by two H-point complex transforms,  a[2m] = IDFT_H(A_e)[m], A_e(k) = A(k) (k = 0: A(0) + A(H)),
// a[2m+1] = IDFT_H(A_o)[m], A_o(k) = A(k) z_k (k = 0: A(0) - A(H)) - the same two transforms per row as K2 runs (there:
// forward of the packed row, inverse of the packed Hilbert spectrum), both register -> register here.  Envelope =
// |a|, then the low-pass in the register layout the transforms leave (f2_envelope_core.h), float64 rows out.
//
// Accuracy guard: the two terms of Y are rounded in float32 separately, each ~6e-8 of the LARGER of (steady response,
// ringing). A row whose ringing after sample n dwarfs everything inside [0, n) (an isolated click in the last
// milliseconds) would therefore lose accuracy relative to its own maximum. Every workgroup computes all M samples
// anyway: the real part of a over the padding region [n, M) - the zero-padded row itself, zero in exact arithmetic
// (the imaginary part is not: a Hilbert transform is not time-limited) - measures the error directly;
// if its maximum exceeds `tol` x the row's maximum the utterance is flagged and the caller's K1 -> K2 launches (which skip
// unflagged utterances) recompute it. Utterances with fewer padding samples than the slowest channel's ringing needs to
// reach its peak (f2_spectral_supports_len) are never routed here.
//
// Traffic: 8 bytes written per sample-channel (the ENV1 rows); tables (1 MB per XCD for 128 channels), utterance
// spectra (64 KB per utterance, shared by its C rows) and 32 bytes of digits per row come from L2. Bound: f32 VALU +
// LDS exchanges.
#define f2fft f2fft_sp   // own copy of the FFT templates (the radix plan of this kernel is chosen independently of K2's)
#include <algorithm>
#include <cmath>
#include <complex>
#include <thread>

#include "f2_fft_lds.h"

using namespace f2fft;

namespace {

typedef float f2_f4 __attribute__((ext_vector_type(4)));
#ifndef F2_SPEC_GB
#define F2_SPEC_GB 16     // bins per load group of the spectrum phase: all 32 loads of a thread in flight (4, 8: measured slower)
#endif

// Diagnostic build only (-DF2_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at the phase boundaries.
#ifdef F2_STAMPS
#define F2_SSTAMP(k)                                         \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        sst[k] = __builtin_amdgcn_s_memrealtime();             \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#else
#define F2_SSTAMP(k) \
    do {             \
    } while (0)
#endif

// ---- row stores through a buffer descriptor with an explicit cache policy ----
// The envelopes are written once and never read by this kernel, 128 KB per row. Stored with the default policy (or
// "nt") the lines stay in the XCD's L2 until evicted and push out the tables every workgroup of the XCD re-reads (2 MB of
// channel tables per XCD + the utterance spectra against 4 MB of L2: measured, half of the table reads then miss L2 and
// the spectrum phase waits on the fabric). sc1 = write through and drop the line (MI355X_MICROARCH.md, store flavours).
#ifndef F2_KS_CGROUP
#define F2_KS_CGROUP 32
#endif
#ifndef F2_KS_CGROUP_LONG
#define F2_KS_CGROUP_LONG 8     // long rows: one 512 KB channel table per XCD at a time
#endif
#ifndef F2_KS_STORE_AUX
#define F2_KS_STORE_AUX 2    // raw buffer store cache policy bits: 0 default, 1 sc0, 2 nt, 16 sc1 (measured: nt 6.74, sc1 6.87, default 7.10 ms)
#endif
typedef unsigned int f2_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int f2_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_buffer(double* y, int n) {
    return __builtin_amdgcn_make_buffer_rsrc(y, 0, n * 8, 0x00020000);
}
// the same from values the compiler may have moved to vector registers (scalar-register pressure): made wave-uniform
// again explicitly - a descriptor held in vector registers turns every buffer access into a readfirstlane loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_buffer_uniform(double* y, int n) {
    const unsigned long long a = (unsigned long long)y;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return row_buffer((double*)(((unsigned long long)hi << 32) | lo), __builtin_amdgcn_readfirstlane(n));
}
// envelope samples i0, i0 + 1, both known to lie inside the row
__device__ __forceinline__ void store_row_pair_inside(__amdgpu_buffer_rsrc_t r, int i0, double a, double b) {
    const f2_d2 v = {a, b};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f2_u4, v), r, i0 * 8, 0, F2_KS_STORE_AUX);
}
// envelope samples i0, i0 + 1 of a row of n samples
__device__ __forceinline__ void store_row_pair_buf(__amdgpu_buffer_rsrc_t r, int n, int i0, double a, double b) {
    if (i0 + 1 < n) {
        const f2_d2 v = {a, b};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(f2_u4, v), r, i0 * 8, 0, F2_KS_STORE_AUX);
    } else if (i0 < n) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(f2_u2, a), r, i0 * 8, 0, F2_KS_STORE_AUX);
    }
}

// ---- first-order low-pass in the register layout of the transforms, constants from a table ----
// Same decomposition as lowpass_pairs_store (f2_envelope_core.h: pair -> weighted DPP scan over the wave -> wave totals ->
// block chain) for NBLK = 16 blocks of float pairs, rearranged for fewer vector instructions per row: every power of
// q = -a1 a thread needs comes from a table built once per (cutoff, workgroup size) on the host instead of float64
// square-and-multiply loops per row; each scan step is ONE v_fmac_f32 with the DPP lane move as its operand (the
// compiler leaves v_mov_b32_dpp + v_fmac_f32 pairs behind); the chain over the wave totals AND over the blocks is run
// once by the first sixteen threads (float64, a DPP scan over the sixteen block totals) instead of by every thread,
// and the carries reach the others as floats.
struct LowpassConsts {
    float qf, b0f;            // q = -a1, b0
    float g1, g2, g4, g8;     // q^2, q^4, q^8, q^16: multipliers of the row_shr 1/2/4/8 steps
    double gw, gblk;          // q^128 (one wave of pairs), q^(2 NT) (one block)
};

template <int NT, int NBLK>
constexpr size_t lowpass_tab_lds_bytes() {
    return sizeof(float) * ((size_t)NBLK * NT + 2 * NBLK * (NT / 64) + NBLK);
}

// one step of the weighted scan on all blocks: sc[j] += g * lane_move(sc[j])
#define F2_DPP_STEP(CTRL)                                                                                                    \
    asm volatile("s_nop 1\n\t"                                                                                               \
                 "v_fmac_f32_dpp %0, %0, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %1, %1, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %2, %2, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %3, %3, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %4, %4, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %5, %5, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %6, %6, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %7, %7, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %8, %8, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %9, %9, %16 " CTRL "\n\t"                                                                   \
                 "v_fmac_f32_dpp %10, %10, %16 " CTRL "\n\t"                                                                 \
                 "v_fmac_f32_dpp %11, %11, %16 " CTRL "\n\t"                                                                 \
                 "v_fmac_f32_dpp %12, %12, %16 " CTRL "\n\t"                                                                 \
                 "v_fmac_f32_dpp %13, %13, %16 " CTRL "\n\t"                                                                 \
                 "v_fmac_f32_dpp %14, %14, %16 " CTRL "\n\t"                                                                 \
                 "v_fmac_f32_dpp %15, %15, %16 " CTRL                                                                        \
                 : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sc[4]), "+v"(sc[5]), "+v"(sc[6]), "+v"(sc[7]),   \
                   "+v"(sc[8]), "+v"(sc[9]), "+v"(sc[10]), "+v"(sc[11]), "+v"(sc[12]), "+v"(sc[13]), "+v"(sc[14]), "+v"(sc[15]) \
                 : "v"(mult))

// Thread t holds the envelope pairs (2m, 2m+1), m = t + NT jj (er[jj], ei[jj]); y[n] = q y[n-1] + b0 (e[n] + e[n-1]) from
// zero state; float64 rows out as coalesced 16-byte stores. lptab[t] = {q^(2 (lane+1)), q^(2 t), (q^2)^((lane & 15) + 1),
// (q^2)^((lane & 31) + 1)}. All threads call it, after a barrier that makes `smem` free.
// CHAINED (rows longer than one sweep of the workgroup, k_spectral_envelope_long): the pairs are those of the samples from
// `ibase` on, e_in = e[ibase - 1], *ychain = y[ibase - 1] on entry and y[ibase + 2 NT NBLK - 1] on return.
// Returns this thread's maximum of |y| over the odd samples it stored (the accuracy guard's denominator when the low-pass is on).
// INSIDE15: every block but the last lies entirely inside the row (n >= 15/16 of the samples the blocks cover): those blocks store
// and take their maxima without a test per lane.
template <int NT, int NBLK, bool CHAINED = false, bool INSIDE15 = false>
__device__ __forceinline__ float lowpass_pairs_store_tab(const float (&er)[NBLK], const float (&ei)[NBLK], const LowpassConsts& K,
                                                        const f2_f4* __restrict__ lptab, unsigned char* smem,
                                                        double* __restrict__ y, int n, int tid, int ibase = 0, float e_in = 0.f,
                                                        double* ychain = nullptr) {
    static_assert(NBLK == 16, "sixteen blocks: one DPP row chains them");
    constexpr int NW = NT / 64;
    float* e1s = reinterpret_cast<float*>(smem);   // [NBLK][NT] odd samples, for e[n-1]
    float* wtot = e1s + NBLK * NT;                 // [NBLK][NW] zero-state value at the end of each wave
    float* cwl = wtot + NBLK * NW;                 // [NBLK][NW] zero-state value of the block entering each wave
    float* ycar = cwl + NBLK * NW;                 // [NBLK] true y entering each block
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) e1s[jj * NT + tid] = ei[jj];
    const f2_f4 tc = lptab[tid];
    __syncthreads();
    float u0[NBLK], u1[NBLK], sc[NBLK];
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) {
        const float eprev = tid > 0 ? e1s[jj * NT + tid - 1] : (jj > 0 ? e1s[(jj - 1) * NT + NT - 1] : (CHAINED ? e_in : 0.f));
        u0[jj] = K.b0f * (er[jj] + eprev);
        u1[jj] = K.b0f * (ei[jj] + er[jj]);
        sc[jj] = fmaf(K.qf, u0[jj], u1[jj]);
    }
    {
        float mult = K.g1;
        F2_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        mult = K.g2;
        F2_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        mult = K.g4;
        F2_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        mult = K.g8;
        F2_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        mult = tc.z;
        F2_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf");
        mult = tc.w;
        F2_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf");
    }
    if (lane == 63) {
#pragma unroll
        for (int jj = 0; jj < NBLK; ++jj) wtot[jj * NW + wv] = sc[jj];
    }
    __syncthreads();
    if (tid < NBLK) {
        // thread jj chains the NW wave totals of block jj (float64), then the sixteen block totals are chained by a
        // weighted scan over the sixteen lanes: ycar[jj] = true y at the end of block jj - 1
        double c = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
            cwl[tid * NW + w2] = (float)c;
            c = fma(K.gw, c, (double)wtot[tid * NW + w2]);
        }
        const double G1 = K.gblk, G2 = G1 * G1, G4 = G2 * G2, G8 = G4 * G4;
        double yin = 0.0;
        if constexpr (CHAINED) {
            yin = *ychain;
            if (tid == 0) c = fma(G1, yin, c);       // the state entering block 0 decays through it
        }
        c = fma(G1, dpp_mov<0x111, 0xF>(c), c);
        c = fma(G2, dpp_mov<0x112, 0xF>(c), c);
        c = fma(G4, dpp_mov<0x114, 0xF>(c), c);
        c = fma(G8, dpp_mov<0x118, 0xF>(c), c);
        const double cprev = dpp_mov<0x111, 0xF>(c);   // (lane 0 reads 0)
        if constexpr (CHAINED) {
            ycar[tid] = tid == 0 ? (float)yin : (float)cprev;
            if (tid == NBLK - 1) *ychain = c;        // (same wave as the reads of *ychain above: program order)
        } else {
            ycar[tid] = (float)cprev;
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t yb = CHAINED ? row_buffer_uniform(y, n) : row_buffer(y, n);
    float ymax = 0.f;
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) {
        const float cw = cwl[jj * NW + wv];
        const float sin_ = fmaf(tc.x, cw, sc[jj]);            // zero-state value at the end of this pair
        const float up = dpp_mov<0x138, 0xF>(sin_);            // wave_shr:1
        const float sprev = lane > 0 ? up : cw;                // ... at the end of the previous pair
        const float y0 = fmaf(K.qf, fmaf(tc.y, ycar[jj], sprev), u0[jj]);
        const float y1 = fmaf(K.qf, y0, u1[jj]);
        const int i0 = ibase + 2 * (tid + NT * jj);
        // every second output sample is enough for the maximum of a low-passed row (it only scales the guard's threshold);
        // block boundaries are wave-uniform: only the block that holds sample n - 1 masks per lane
        if constexpr (INSIDE15) {
            if (jj < NBLK - 1) {
                store_row_pair_inside(yb, i0, (double)y0, (double)y1);
                ymax = fmaxf(ymax, fabsf(y1));
            } else {
                store_row_pair_buf(yb, n, i0, (double)y0, (double)y1);
                ymax = fmaxf(ymax, i0 + 1 < n ? fabsf(y1) : (i0 < n ? fabsf(y0) : 0.f));
            }
        } else if constexpr (CHAINED) {
            // (the long-row kernel is short of registers: per-lane tests cost it less than the uniform branches' live ranges)
            store_row_pair_buf(yb, n, i0, (double)y0, (double)y1);
            const int lo = ibase + 2 * NT * jj;
            if (lo + 2 * NT <= n) ymax = fmaxf(ymax, fabsf(y1));
            else if (lo < n) ymax = fmaxf(ymax, i0 + 1 < n ? fabsf(y1) : fabsf(y0));
        } else {
            const int lo = ibase + 2 * NT * jj;
            if (lo + 2 * NT <= n) {                      // (wave-uniform: the whole block lies inside the row)
                store_row_pair_inside(yb, i0, (double)y0, (double)y1);
                ymax = fmaxf(ymax, fabsf(y1));
            } else if (lo < n) {
                store_row_pair_buf(yb, n, i0, (double)y0, (double)y1);
                ymax = fmaxf(ymax, i0 + 1 < n ? fabsf(y1) : (i0 < n ? fabsf(y0) : 0.f));
            }
        }
    }
    return ymax;
}

// (the read-only tables are separate __restrict__ kernel arguments: pointers inside a by-value struct carry no
// no-alias information, and the wave-uniform reads among them would then go through the vector memory path)
struct SpecParams {
    int64_t xpitch;           // X: [nutt][xpitch] spectra of the utterances of this launch, k = 0..H
    int64_t tpitch;           // HU: [C][tpitch] {Hs.re, Hs.im, u.re, u.im}, Hs = (2/M) s N_1..N_4 u^4
    double* env;
    int* uflag;               // [B] set to 1 when a row of the utterance fails the accuracy guard
    int C;
    int nutt;
    int lpf;
    LowpassConsts lp;
    float tol;
    float* gdump;             // diagnostic (option "spectral_guard_dump"): [B][C][4] {row max, padding residual, low-passed row max, flagged}
    unsigned long long* stamps;   // diagnostic build only
};

// Y'(k) = (2/M) Y(k) of one bin
__device__ __forceinline__ cpx<float> spectral_bin(cpx<float> X, f2_f4 hu, cpx<float> w, cpx<float> z, const float* __restrict__ rho) {
    const float ur = hu.z, ui = hu.w;
    float tr = fmaf(rho[1], w.re, rho[0]), ti = rho[1] * w.im;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const float rr = fmaf(rho[2 * j + 1], w.re, rho[2 * j]), ri = rho[2 * j + 1] * w.im;
        const float nr = fmaf(-ui, ti, fmaf(ur, tr, rr));
        const float ni = fmaf(ui, tr, fmaf(ur, ti, ri));
        tr = nr;
        ti = ni;
    }
    const float qr = ur * tr - ui * ti, qi = ur * ti + ui * tr;
    float yr = hu.x * X.re;
    yr = fmaf(-hu.y, X.im, yr);
    yr = fmaf(z.re, qr, yr);
    yr = fmaf(-z.im, qi, yr);
    float yi = hu.x * X.im;
    yi = fmaf(hu.y, X.re, yi);
    yi = fmaf(z.re, qi, yi);
    yi = fmaf(z.im, qr, yi);
    return {yr, yi};
}

#include "f2_fft13_merged.h"

// w_k of bin k = tid + j NB0: w_tid exp(-2 pi i j / (2 R0))  (NB0 / M = 1 / (2 R0))
template <int R0, int J = 0>
__device__ __forceinline__ cpx<float> bin_w(cpx<float> w0, int j) {
    if constexpr (J < R0) {
        if (j == J) return mulw<2 * R0, J>(w0);
        return bin_w<R0, J + 1>(w0, j);
    } else {
        return w0;
    }
}

// maximum over the wave of non-negative values (two at once: independent chains), results valid in lane 63: row_shr 1/2/4/8, row_bcast 15/31,
// every step ONE v_max_f32 with the lane move as its operand (the compiler leaves v_mov_b32_dpp + v_max_f32 pairs; lanes
// without a source lane read 0, rows outside the mask keep their value)
#define F2_DPP_MAX2(CTRL)                                              \
    asm volatile("s_nop 1\n\t"                                         \
                 "v_max_f32_dpp %0, %0, %0 " CTRL "\n\t"               \
                 "v_max_f32_dpp %1, %1, %1 " CTRL                       \
                 : "+v"(a), "+v"(b))
__device__ __forceinline__ void wave_max63x2(float& a, float& b) {
    F2_DPP_MAX2("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    F2_DPP_MAX2("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    F2_DPP_MAX2("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    F2_DPP_MAX2("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
    F2_DPP_MAX2("row_bcast:15 row_mask:0xa bank_mask:0xf");
    F2_DPP_MAX2("row_bcast:31 row_mask:0xc bank_mask:0xf");
}
__device__ __forceinline__ float wave_max63(float v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return v;
}

// max |Re a| over this thread's samples in the padding region [n, M): v[brev(j)] = sample 2 (tid + NT j) + odd.
// PADLAST: every row of the launch has its padding inside the last block (n >= 15/16 M: every 1 s row of a 16384-point class).
template <int R0, int NT, bool PADLAST>
__device__ __forceinline__ float pad_residual(const cpx<float> (&v)[R0], int n, int tid, int odd, float gout) {
    if constexpr (PADLAST) {
        return fmaxf(gout, 2 * (tid + NT * (R0 - 1)) + odd >= n ? fabsf(v[brev<R0>(R0 - 1)].re) : 0.f);
    } else {
        // sample 2 (tid + NT j) + odd >= n  <=>  j >= jmin (one register; the compiler would otherwise keep sixteen sample indices)
        int jmin = (n - 2 * tid - odd + 2 * NT - 1) >> (__builtin_ctz(2 * NT));
        asm volatile("" : "+v"(jmin));
        // (rows of this length class have n > M / 2: the first half of the blocks never holds padding)
#pragma unroll
        for (int j = R0 / 2; j < R0; ++j) gout = fmaxf(gout, j >= jmin ? fabsf(v[brev<R0>(j)].re) : 0.f);
        return gout;
    }
}

// one thread per row, after the barrier behind the last atomicMax: flag the utterance if the padding-region residual exceeds
// tol x the maximum of the row AS DELIVERED (low-passed when the low-pass is on)
__device__ __forceinline__ void guard_decide(const SpecParams& P, const unsigned* guard, int b, int c) {
    const float gi = __uint_as_float(guard[0]), go = __uint_as_float(guard[1]), gl = __uint_as_float(guard[2]);
    const float den = P.lpf ? gl : gi;
    const bool flag = go > P.tol * den;
    if (flag) P.uflag[b] = 1;
    if (P.gdump) {
        float* g = P.gdump + ((size_t)b * (size_t)P.C + (size_t)c) * 4;
        g[0] = gi;
        g[1] = go;
        g[2] = gl;
        g[3] = flag ? 1.f : 0.f;
    }
}

template <int LOG2H, bool PADLAST>
__global__ __launch_bounds__((threads_for<float, LOG2H>()), (min_waves_for<float, LOG2H>())) void k_spectral_envelope(
    SpecParams P, const cpx<float>* __restrict__ Xall /* utterance spectra */, const f2_f4* __restrict__ HUall,
    const cpx<float>* __restrict__ E /* [M] exp(-2 pi i q / M) */, const float* __restrict__ rho_all /* [rows][8] digits of Q */,
    const int64_t* __restrict__ offsets, const int* __restrict__ ulist, const f2_f4* __restrict__ lptab,
    const cpx<float>* __restrict__ tw) {
    constexpr int NT = threads_for<float, LOG2H>();
    constexpr int H = 1 << LOG2H;
    constexpr int M = 2 * H;
    constexpr int CS = cpad_size(H);
    constexpr int R0 = 1 << plan_bits(LOG2H, 0);
    constexpr int NB0 = H / R0;
    static_assert(NB0 % NT == 0, "every thread owns whole butterflies of the first pass");
    constexpr int ITER0 = NB0 / NT;
    constexpr int PT = plan_points_per_thread(LOG2H, NT);
    constexpr int NBLK = ITER0 * R0;
    static_assert(PT == NBLK, "register shape");
    constexpr size_t LP = lowpass_tab_lds_bytes<NT, NBLK>();
    constexpr int LDS_BYTES = (int)((size_t)CS * 8 > LP ? (size_t)CS * 8 : LP);
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    constexpr int TWL = plan_tw_lds_count(LOG2H);
    __shared__ __attribute__((aligned(16))) cpx<float> twl[TWL > 0 ? TWL : 1];
    __shared__ unsigned guard[4];   // max |a| inside [0, n) / Re a inside [n, M) / low-passed row maximum, as float bits (>= 0); waves counted in
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem);

    const int tid = threadIdx.x;
    // Rows in the order (channel group of CG channels, utterance, channel in the group): the workgroups in flight then
    // share CG channel tables instead of C (workgroup b runs on XCD b % 8: CG / 8 tables of 128 KB per XCD's L2) while an
    // utterance's spectrum is still used by CG consecutive workgroups. Measured on 1000 x 1 s, 128 channels: 6.54 ms
    // with CG = 32 against 6.71 ms with the plain (utterance, channel) order.
    int u, c;
    {
        constexpr int CG = F2_KS_CGROUP;
        const int nfull = P.C / CG, per = P.nutt * CG;
        const int cg = min((int)(blockIdx.x / per), nfull);          // the last group may be partial
        const int rr = blockIdx.x - cg * per;
        const int gsz = cg < nfull ? CG : P.C - nfull * CG;
        u = rr / gsz;
        c = cg * CG + (rr - u * gsz);
    }
    const int row_id = u * P.C + c;
    const int b = ulist[u];
    const int64_t off = offsets[b];
    const int n = (int)(offsets[b + 1] - off);
    double* __restrict__ y = P.env + ((size_t)P.C * (size_t)off + (size_t)c * (size_t)n);
    const float* __restrict__ rho = rho_all + (size_t)row_id * 8;
#ifdef F2_KS_HOT_TABLES   // knock-out (timing only): every row reads utterance 0's spectrum and channel 0's table
    const cpx<float>* __restrict__ Xu = Xall;
    const f2_f4* __restrict__ HUc = HUall;
#else
    const cpx<float>* __restrict__ Xu = Xall + (size_t)u * P.xpitch;
    const f2_f4* __restrict__ HUc = HUall + (size_t)c * P.tpitch;
#endif

#ifdef F2_STAMPS
    unsigned long long sst[8] = {0};
#endif
    F2_SSTAMP(0);
    for (int i = tid; i < TWL; i += NT) twl[i] = tw[plan_tw_offset(LOG2H, 1) + i];
    if (tid < 4) guard[tid] = 0u;

    // 1. spectrum of the zero-padded row at this thread's bins k = tid + j NB0 (first-pass register layout). Per bin two
    //    loads (utterance spectrum, channel table); the two phase factors follow from one load each: w_k = w_tid e^{-2 pi i
    //    j / 32} (compile-time constants) and w_k^n = w_tid^n (w_NB0^n)^j (wave-uniform factors, scalar loads). The bins
    //    go in groups of GB with the next group's loads in flight - all 32 loads at once would need 96 registers.
    static_assert(ITER0 == 1 && R0 == 16, "one radix-16 butterfly per thread in the first pass");
    constexpr int GB = F2_SPEC_GB, NG = R0 / GB;
    cpx<float> yk[PT], v[PT];
    // rows whose first pass is one radix-16 butterfly per thread (all three length classes served): the two transforms share it
    static_assert(NB0 == NT && plan_npass(LOG2H) >= 3, "fft_pass0_pair + fft_from_pass0");
    cpx<float> vo[PT];   // conj(A_o) / M = conj(A) w_k / M, formed while the bin's phase factor is at hand
    const unsigned zstep = ((unsigned)NB0 * (unsigned)n) & (M - 1);
    cpx<float> w0 = E[tid];
    cpx<float> z0 = E[__umul24((unsigned)tid, (unsigned)n) & (M - 1)];
    const cpx<float> XH = Xu[H];     // Nyquist bin (wave-uniform addresses: scalar loads, issued up front)
    const f2_f4 HUH = HUc[H];
    cpx<float> Xl[2][GB];
    f2_f4 Hl[2][GB];
    // `kb` is laundered through an empty asm together with a result of each group, so that the loads of group g + 2
    // cannot be issued before group g has been computed (the compiler would otherwise hoist all 32 loads and spill)
    //  (unsigned lane offset + wave-uniform row pointers: one offset register serves every load)
    unsigned kb = (unsigned)tid;
#ifdef F2_KS_KO_LOADS   // knock-out (timing only, results wrong): table values from arithmetic instead of memory
#define F2_KS_LOADX(ptr, idx) cpx<float>{1.0f + 1e-3f * (float)(idx), 1e-3f * (float)(idx)}
#define F2_KS_LOADH(ptr, idx) f2_f4{1e-3f, 1e-4f * (float)(idx), 0.5f, 1e-3f * (float)(idx)}
#else
#define F2_KS_LOADX(ptr, idx) (ptr)[idx]
#define F2_KS_LOADH(ptr, idx) (ptr)[idx]
#endif
#pragma unroll
    for (int q = 0; q < GB; ++q) {
        Xl[0][q] = F2_KS_LOADX(Xu + q * NB0, kb);
        Hl[0][q] = F2_KS_LOADH(HUc + q * NB0, kb);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                Xl[(g + 1) & 1][q] = F2_KS_LOADX(Xu + ((g + 1) * GB + q) * NB0, kb);
                Hl[(g + 1) & 1][q] = F2_KS_LOADH(HUc + ((g + 1) * GB + q) * NB0, kb);
            }
        }
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            const int j = g * GB + q;
            const cpx<float> zj = E[(j * zstep) & (M - 1)];                       // wave-uniform
            const cpx<float> wj = bin_w<R0>(w0, j);
            yk[j] = spectral_bin(Xl[g & 1][q], Hl[g & 1][q], wj, cmul(z0, zj), rho);
            vo[j] = {yk[j].re * wj.re + yk[j].im * wj.im, yk[j].re * wj.im - yk[j].im * wj.re};
#ifdef F2_STAMPS
            if (j == 0 || j == 7) {
                asm volatile("s_nop 0" : "+v"(yk[j].re), "+v"(yk[j].im));
                sst[j == 0 ? 1 : 2] = __builtin_amdgcn_s_memrealtime();
                asm volatile("s_nop 0" : "+v"(yk[j].re));
            }
#endif
        }
        // (w0 / z0 pass through as well: the phase factors of a later group are then not formed ahead of time either)
        if constexpr (GB == 4) {
            asm volatile(""
                         : "+v"(kb), "+v"(w0.re), "+v"(w0.im), "+v"(z0.re), "+v"(z0.im), "+v"(yk[g * GB].re), "+v"(yk[g * GB + 1].re),
                           "+v"(yk[g * GB + 2].re), "+v"(yk[g * GB + 3].re));
        } else {
            asm volatile("" : "+v"(kb), "+v"(w0.re), "+v"(w0.im), "+v"(z0.re), "+v"(z0.im), "+v"(yk[g * GB].re), "+v"(yk[g * GB + GB - 1].re));
        }
    }
#ifdef F2_STAMPS
    asm volatile("s_nop 0" : "+v"(yk[15].re), "+v"(yk[15].im));
    sst[3] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 0" : "+v"(yk[15].re));
#endif
    // k = 0 carries the Nyquist term: A_e(0) = A(0) + A(H), A_o(0) = A(0) - A(H) (both real); thread 0 keeps
    // (Y'(0), Y'(H)) in yk[0]
    if (tid == 0) {
        const cpx<float> yh = spectral_bin(XH, HUH, cpx<float>{-1.f, 0.f}, cpx<float>{(n & 1) ? -1.f : 1.f, 0.f}, rho);
        yk[0] = {yk[0].re, yh.re};
    }
    float er[NBLK], ei[NBLK];
    // 4. accuracy guard: block jj of this thread is the sample pair 2 (tid + NT jj), + 1. Inside [0, n) the row's maximum;
    //    over the WHOLE padding region [n, M) the real part of a (the zero-padded row itself: zero in exact arithmetic,
    //    while the imaginary part, the Hilbert transform of a time-limited signal, is not) - the error itself, including the
    //    ringing-shaped part that starts at n. Block boundaries are wave-uniform: only blocks that hold padding pay for it.
    float gin = 0.f, gout = 0.f;
    constexpr bool T0R = derive_tw0<float, LOG2H>();
    {
        // 2 + 3: both transforms' inputs formed at once - conj(A_e) / M and conj(A_o) / M =
        // conj(A) w_k / M - and their first passes run together (one set of derived twiddles); the odd transform's
        // first-pass outputs then wait in the registers Y' occupied, while the even transform goes through LDS
#pragma unroll
        for (int j = 0; j < R0; ++j) v[j] = {yk[j].re, -yk[j].im};
        if (tid == 0) {
            v[0] = {0.5f * (yk[0].re + yk[0].im), 0.f};
            vo[0] = {0.5f * (yk[0].re - yk[0].im), 0.f};
        }
        F2_SSTAMP(4);
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e), "+v"(v[0].re), "+v"(vo[PT - 1].im));
        fft_pass0_pair<LOG2H, NT, PT>(tw, tid_e, v, vo);
        fft_from_pass0<LOG2H, PT, NT, T0R>(lds, tw, twl, tid_e, v);
        gout = pad_residual<R0, NT, PADLAST>(v, n, tid, 0, gout);
#pragma unroll
        for (int j = 0; j < R0; ++j) {
            const cpx<float> a = v[brev<R0>(j)];
            er[j] = fsqrt(a.re * a.re + a.im * a.im);
        }
        fft_from_pass0<LOG2H, PT, NT, T0R>(lds, tw, twl, tid_e, vo);
        F2_SSTAMP(5);
        gout = pad_residual<R0, NT, PADLAST>(vo, n, tid, 1, gout);
#pragma unroll
        for (int j = 0; j < R0; ++j) {
            const cpx<float> a = vo[brev<R0>(j)];
            ei[j] = fsqrt(a.re * a.re + a.im * a.im);
        }
    }
#pragma unroll
    for (int jj = 0; jj < NBLK; ++jj) {
        const int lo = 2 * NT * jj;
        if ((PADLAST && jj < NBLK - 1) || lo + 2 * NT <= n) {     // (PADLAST: every block but the last is inside the row)
            gin = fmaxf(gin, fmaxf(er[jj], ei[jj]));
        } else if (lo < n) {
            const int i0 = lo + 2 * tid;
            gin = fmaxf(gin, fmaxf(i0 < n ? er[jj] : 0.f, i0 + 1 < n ? ei[jj] : 0.f));
        }
    }
    float glp = 0.f;
    wave_max63x2(gin, gout);
    if ((tid & 63) == 63) {           // (posted here: they travel while the low-pass runs)
        __hip_atomic_fetch_max(&guard[0], __float_as_uint(gin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(&guard[1], __float_as_uint(gout), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    F2_SSTAMP(6);
    // 5. stores (the last pass of the transform ended with a barrier after its LDS reads: smem is free)
    if (!P.lpf) {
        const __amdgpu_buffer_rsrc_t yb = row_buffer(y, n);
#pragma unroll
        for (int jj = 0; jj < NBLK; ++jj) {
            if ((PADLAST && jj < NBLK - 1) || 2 * NT * (jj + 1) <= n) store_row_pair_inside(yb, 2 * (tid + NT * jj), (double)er[jj], (double)ei[jj]);
            else if (2 * NT * jj < n) store_row_pair_buf(yb, n, 2 * (tid + NT * jj), (double)er[jj], (double)ei[jj]);
        }
    } else {
        // the parity bar is written on the LOW-PASSED row (EnvelopeExtraction.py:57-66): its maximum, which a bursty row
        // keeps several times below the raw one, is what the residual is compared with
        glp = wave_max63(lowpass_pairs_store_tab<NT, NBLK, false, PADLAST>(er, ei, P.lp, lptab, smem, y, n, tid));   // (contains barriers)
    }
    // No barrier for the verdict: every wave posts its maxima and then counts itself in; the LDS serves a wave's operations in
    // order, so the wave that counts in last sees everybody's maxima and decides for the row.
    if ((tid & 63) == 63) {
        if (P.lpf) __hip_atomic_fetch_max(&guard[2], __float_as_uint(glp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_fetch_add(&guard[3], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(NT / 64 - 1))
            guard_decide(P, guard, b, c);
    }
#ifdef F2_STAMPS
    F2_SSTAMP(7);
    if (tid == 0 && P.stamps)
        for (int k = 0; k < 8; ++k) P.stamps[(size_t)blockIdx.x * 8 + k] = sst[k];
#endif
}

// ---- rows of 32769 .. 65472 samples: M = 65536, H = 32768 ----
// The two H-point transforms of such a row (131072 + 131072 bytes as float32) do not fit the LDS of a CU. Decimating
// the output once more, a[4 m + r] = IDFT_Q(B_r)[m], Q = H / 2 = 16384, r = 0..3, with the folded one-sided spectrum
//     conj B_r(k) = w_k^r [ conj A'(k) + (-i)^r conj A'(k + Q) ],   k < Q,   A' = A / M  (k = 0 also takes (-1)^r A'(H)),
// gives four 16384-point transforms per row, run as two passes of the pair (r, r + 2) = (p, p + 2): the pair shares
// P = conj A'(k), R = (-i)^p conj A'(k + Q) (v = w^p (P + R), vo = w^(p+2) (P - R)) and its first transform pass, exactly
// like the even / odd pair of the shorter rows - the spectrum bins are formed twice per row (once per pass) instead of
// being kept (32 bins = 64 registers per thread). The 4 x 16 magnitudes per thread cannot wait in registers for the
// low-pass either (a first version that kept them spilled 150 registers and ran at 195 us per row): every pass parks
// its magnitudes as float32 in the upper half of the row's own output slot (bytes 4 n + 4 i of the 8 n: sample i of
// the envelope lands at byte 8 i, so the two sweeps of the low-pass below - samples [0, 32768) and [32768, n) - each read
// their parked inputs before any of their outputs could overwrite them, barriers in between; the parked lines stay in
// L2 / the Infinity Cache: 4 bytes written + read per sample next to the 8 stored).
// Parking layout (float index inside the row's 2 n floats): two banks - samples below / from 32768, i.e. the inputs of
// the first / second low-pass sweep - at n and n + 32768; inside a bank one plane per r (a transform's outputs are then
// stored as whole 256-byte runs per wave instead of one float every 16 bytes), plane sizes summing to the bank's samples.
// Guard, flags, row order and arguments as k_spectral_envelope; `tw` = the tables of the 16384-point transform.
#ifndef F2_KSL_PARK_AUX
#define F2_KSL_PARK_AUX 0    // cache policy of the parked magnitudes (stores and loads)
#endif
// float index of plane r of bank `bank` (m counted from the bank's first sample)
__device__ __forceinline__ int park_plane(int n, int bank, int r) {
    constexpr int MB = 8192;                      // m of the first sample of bank 1
    if (bank == 0) return n + r * MB;
    const int nb = n - 4 * MB;                    // samples in bank 1; planes q = 0..r-1 hold ceil((nb - q) / 4) each
    const int before = r == 0 ? 0 : r == 1 ? (nb + 3) >> 2 : r == 2 ? ((nb + 3) >> 2) + ((nb + 2) >> 2) : nb - (nb >> 2);
    return n + 4 * MB + before;
}
__global__ __launch_bounds__(1024, 4) void k_spectral_envelope_long(
    SpecParams P, const cpx<float>* __restrict__ Xall, const f2_f4* __restrict__ HUall, const cpx<float>* __restrict__ E,
    const float* __restrict__ rho_all, const int64_t* __restrict__ offsets, const int* __restrict__ ulist,
    const f2_f4* __restrict__ lptab, const cpx<float>* __restrict__ tw) {
    constexpr int LOG2Q = 14, NT = 1024, Q = 1 << LOG2Q, H = 2 * Q, M = 2 * H;
    static_assert(threads_for<float, LOG2Q>() == NT, "the 16-16-4-16 plan on 1024 threads");
    constexpr int CS = cpad_size(Q);
    constexpr int R0 = 16, NB0 = Q / R0, PT = 16, NBLK = 16;
    static_assert(NB0 == NT && plan_points_per_thread(LOG2Q, NT) == PT, "one radix-16 butterfly per thread in the first pass");
    constexpr size_t LP = lowpass_tab_lds_bytes<NT, NBLK>();
    constexpr int LDS_BYTES = (int)((size_t)CS * 8 > LP ? (size_t)CS * 8 : LP);
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    constexpr int TWL = plan_tw_lds_count(LOG2Q);
    __shared__ __attribute__((aligned(16))) cpx<float> twl[TWL];
    __shared__ unsigned guard[4];
    __shared__ double ychain;
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem);

    const int tid = threadIdx.x;
    int u, c;
    {
        constexpr int CG = F2_KS_CGROUP_LONG;
        const int nfull = P.C / CG, per = P.nutt * CG;
        const int cg = min((int)(blockIdx.x / per), nfull);
        const int rr = blockIdx.x - cg * per;
        const int gsz = cg < nfull ? CG : P.C - nfull * CG;
        u = rr / gsz;
        c = cg * CG + (rr - u * gsz);
    }
    const int row_id = u * P.C + c;
    const int b = ulist[u];
    const int64_t off = offsets[b];
    const int n = (int)(offsets[b + 1] - off);
    double* __restrict__ y = P.env + ((size_t)P.C * (size_t)off + (size_t)c * (size_t)n);
    const float* __restrict__ rho = rho_all + (size_t)row_id * 8;
    const cpx<float>* __restrict__ Xu = Xall + (size_t)u * P.xpitch;
    const f2_f4* __restrict__ HUc = HUall + (size_t)c * P.tpitch;

    for (int i = tid; i < TWL; i += NT) twl[i] = tw[plan_tw_offset(LOG2Q, 1) + i];
    if (tid < 4) guard[tid] = 0u;
    if (tid == 0) ychain = 0.0;

    const unsigned zstep = ((unsigned)NB0 * (unsigned)n) & (M - 1);
    cpx<float> w0 = E[tid];
    cpx<float> z0 = E[((unsigned)tid * (unsigned)n) & (M - 1)];
    const cpx<float> zq = E[((unsigned)Q * (unsigned)n) & (M - 1)];   // w_Q^n = (-i)^n
    float yh = 0.f;   // Y'(H) (real)
    if (tid == 0) yh = spectral_bin(Xu[H], HUc[H], cpx<float>{-1.f, 0.f}, cpx<float>{(n & 1) ? -1.f : 1.f, 0.f}, rho).re;
    constexpr bool T0R = derive_tw0<float, LOG2Q>();
    float gin = 0.f, gout = 0.f;
    unsigned kb = (unsigned)tid;
#ifdef F2_STAMPS
    unsigned long long sst[8] = {0};
#endif
    F2_SSTAMP(0);
#pragma unroll 1
    for (int p = 0; p < 2; ++p) {
        cpx<float> v[PT], vo[PT];
        constexpr int GB = 4;
#pragma unroll
        for (int g = 0; g < R0 / GB; ++g) {
            cpx<float> Xl[2][GB];
            f2_f4 Hl[2][GB];
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                Xl[0][q] = F2_KS_LOADX(Xu + (g * GB + q) * NB0, kb);
                Hl[0][q] = F2_KS_LOADH(HUc + (g * GB + q) * NB0, kb);
                Xl[1][q] = F2_KS_LOADX(Xu + Q + (g * GB + q) * NB0, kb);
                Hl[1][q] = F2_KS_LOADH(HUc + Q + (g * GB + q) * NB0, kb);
            }
#pragma unroll
            for (int q = 0; q < GB; ++q) {
                const int j = g * GB + q;
                const cpx<float> w = cmul(w0, E[j * NB0]);                       // w_k, k = tid + j NB0 (wave-uniform factor)
                const cpx<float> z = cmul(z0, E[(j * zstep) & (M - 1)]);          // w_k^n
                const cpx<float> ya = spectral_bin(Xl[0][q], Hl[0][q], w, z, rho);
                const cpx<float> yb2 = spectral_bin(Xl[1][q], Hl[1][q], cpx<float>{w.im, -w.re}, cmul(z, zq), rho);   // bin k + Q
                cpx<float> cp = {ya.re, -ya.im};
                if (j == 0 && tid == 0) cp = {0.5f * (ya.re + (p ? -yh : yh)), 0.f};
                const cpx<float> cr = p ? cpx<float>{-yb2.im, -yb2.re} : cpx<float>{yb2.re, -yb2.im};   // (-i)^p conj
                const cpx<float> sm = cp + cr, df = cp - cr;
                const cpx<float> w2 = cmul(w, w);
                v[j] = p ? cmul(sm, w) : sm;
                vo[j] = cmul(df, p ? cmul(w2, w) : w2);
            }
            // the next group's loads are not issued before this group has been consumed (registers)
            asm volatile("" : "+v"(kb), "+v"(w0.re), "+v"(w0.im), "+v"(z0.re), "+v"(z0.im), "+v"(v[g * GB].re), "+v"(vo[g * GB + GB - 1].im));
        }
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e), "+v"(v[0].re), "+v"(vo[PT - 1].im));
#ifdef F2_STAMPS
        if (p == 0) F2_SSTAMP(1);
        else F2_SSTAMP(3);
#endif
        fft_pass0_pair<LOG2Q, NT, PT>(tw, tid_e, v, vo);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // conj a[4 m + r], m = tid + j NT in (half ? vo : v)[brev(j)], r = p + 2 half
            if (half == 0) fft_from_pass0<LOG2Q, PT, NT, T0R>(lds, tw, twl, tid_e, v);
            else fft_from_pass0<LOG2Q, PT, NT, T0R>(lds, tw, twl, tid_e, vo);
            const int r = p + 2 * half;
            const __amdgpu_buffer_rsrc_t yb = row_buffer_uniform(y, n);
            // (one lane offset per bank; the block j adds a compile-time 4096 bytes)
            const int pv0 = 4 * (park_plane(n, 0, r) + tid), pv1 = 4 * (park_plane(n, 1, r) + tid);
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const cpx<float> a = half ? vo[brev<R0>(j)] : v[brev<R0>(j)];
                const float e = fsqrt(a.re * a.re + a.im * a.im);
                const bool in = 4 * (tid + NT * j) + r < n;
                // (no branch per sample: a sample beyond the row goes to an offset the buffer's range check drops)
#ifndef F2_KSL_KO_PARK   // knock-out (timing only)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e), yb, in ? (j < 8 ? pv0 : pv1) : 0x7ffffff0, 4 * NT * (j & 7),
                                                      F2_KSL_PARK_AUX);
#endif
                gin = fmaxf(gin, in ? e : 0.f);
                // (rows of this class have n > 32768: blocks 0 .. 7 of 4096 samples never hold padding; wave-uniform test for the others)
                if (j >= 8 && 4 * NT * (j + 1) > n) gout = fmaxf(gout, in ? 0.f : fabsf(a.re));
            }
        }
#ifdef F2_STAMPS
        if (p == 0) F2_SSTAMP(2);
        else F2_SSTAMP(4);
#endif
    }
    gin = wave_max63(gin);
    gout = wave_max63(gout);
    if ((tid & 63) == 63) {
        atomicMax(&guard[0], __float_as_uint(gin));
        atomicMax(&guard[1], __float_as_uint(gout));
    }
    // the parked magnitudes are read back by other threads of the same workgroup (same CU, same L1): workgroup scope -
    // an agent-scope fence writes the XCD's whole L2 back (measured: 8 x the kernel time)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // low-pass + stores: two sweeps of 2 NT NBLK = 32768 samples in the pair layout of lowpass_pairs_store_tab
    // (loading the parked inputs of both sweeps up front costs 90 spilled registers next to the low-pass's own 80)
#ifdef F2_KSL_KO_SWEEP   // knock-out (timing only)
    if (n < 0)
#endif
    float glp = 0.f;
#pragma unroll 1
    for (int sw = 0; sw < 2; ++sw) {
        const int ibase = sw * 2 * NT * NBLK;
        const __amdgpu_buffer_rsrc_t yb = row_buffer_uniform(y, n);
        float er[NBLK], ei[NBLK];
        // samples i0 = ibase + 2 (tid + NT jj) = 4 m + r and i0 + 1, r = 0 or 2: the same m of two neighbouring planes of bank sw
        const int rl = 2 * (tid & 1);
        const int pva = 4 * (park_plane(n, sw, rl) + (tid >> 1)), pvb = 4 * (park_plane(n, sw, rl + 1) + (tid >> 1));
#pragma unroll
        for (int jj = 0; jj < NBLK; ++jj) {
            const int i0 = ibase + 2 * (tid + NT * jj);
            // (samples beyond the row: an out-of-range offset, which reads as 0)
            er[jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yb, i0 < n ? pva : 0x7ffffff0, 2 * NT * jj, F2_KSL_PARK_AUX));
            ei[jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yb, i0 + 1 < n ? pvb : 0x7ffffff0, 2 * NT * jj, F2_KSL_PARK_AUX));
        }
        if (!P.lpf) {
            __syncthreads();   // every load of the sweep before any of its stores
#pragma unroll
            for (int jj = 0; jj < NBLK; ++jj) store_row_pair_buf(yb, n, ibase + 2 * (tid + NT * jj), (double)er[jj], (double)ei[jj]);
        } else {
            const float e_in =   // sample 32767: the last of plane 3 of bank 0
                sw ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yb, 4 * (park_plane(n, 0, 3) + 8191), 0, F2_KSL_PARK_AUX)) : 0.f;
            glp = fmaxf(glp, lowpass_pairs_store_tab<NT, NBLK, true>(er, ei, P.lp, lptab, smem, y, n, tid, ibase, e_in, &ychain));
        }
        __syncthreads();   // (the second sweep's inputs were parked before the first barrier above; `smem` is free again)
#ifdef F2_STAMPS
        if (sw == 0) F2_SSTAMP(5);
        else F2_SSTAMP(6);
#endif
    }
    // (verdict without a barrier, as in k_spectral_envelope: the wave that counts itself in last decides)
    if (P.lpf) glp = wave_max63(glp);
    if ((tid & 63) == 63) {
        if (P.lpf) __hip_atomic_fetch_max(&guard[2], __float_as_uint(glp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_fetch_add(&guard[3], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(NT / 64 - 1))
            guard_decide(P, guard, b, c);
    }
#ifdef F2_STAMPS
    F2_SSTAMP(7);
    if (tid == 0 && P.stamps)
        for (int k = 0; k < 8; ++k) P.stamps[(size_t)blockIdx.x * 8 + k] = sst[k];
#endif
}

// ---- X = DFT_M(x zero-padded), k = 0..H, float64 arithmetic, float32 out ----
// (float64: the float32 rounding of a transform is relative to the LOUDEST band of the utterance; a channel that sits
// 60 dB below it - speech above 4 kHz - would inherit 1e-4 of its own level.)
// One workgroup transforms one real sequence of 2 HS samples as HS packed complex points in LDS (HS <= 8192 in
// float64). Utterances padded to M = 2 HS go straight to the float32 table (LOGD = 0). Longer ones (M = 2^LOGD 2 HS) are
// decimated in time: workgroup (utterance, p) transforms x_p[m] = x[D m + p] and leaves its float64 spectrum
// X_p(k), k = 0..HS; k_spectrum_combine then forms X(k) = sum_p W_M^(p k) X_p(k mod 2 HS) (X_p is 2 HS-periodic and
// conjugate-symmetric), k = 0..M/2.
template <typename WaveT, int LOG2HS, int LOGD>
__global__ __launch_bounds__((threads_for<double, LOG2HS>())) void k_utterance_spectrum(const WaveT* __restrict__ wave,
                                                                                      const int64_t* __restrict__ offsets,
                                                                                      const int* __restrict__ ulist,
                                                                                      cpx<float>* __restrict__ X, int64_t xpitch,
                                                                                      cpx<double>* __restrict__ Xpart,
                                                                                      const cpx<double>* __restrict__ tw) {
    constexpr int LOG2H = LOG2HS;
    constexpr int D = 1 << LOGD;
    constexpr int NT = threads_for<double, LOG2H>();
    constexpr int H = 1 << LOG2H;
    constexpr int CS = cpad_size(H);
    constexpr int R0 = 1 << plan_bits(LOG2H, 0);
    constexpr int NB0 = H / R0;
    constexpr int ITER0 = (NB0 + NT - 1) / NT;
    constexpr int PT = plan_points_per_thread(LOG2H, NT);
    constexpr bool FULL0 = NB0 % NT == 0;
    __shared__ __attribute__((aligned(16))) cpx<double> lds[CS];
    constexpr int TWL = plan_tw_lds_count(LOG2H);
    __shared__ __attribute__((aligned(16))) cpx<double> twl[TWL > 0 ? TWL : 1];
    const cpx<double>* __restrict__ V = tw + plan_tw_total(LOG2H);   // exp(-2 pi i k / (2 H)), k <= H/2
    const int tid = threadIdx.x;
    const int ul = blockIdx.x >> LOGD, ph = blockIdx.x & (D - 1);
    const int b = ulist[ul];
    const int64_t off = offsets[b];
    const int n = (int)(offsets[b + 1] - off);
    const WaveT* __restrict__ x = wave + off;
    for (int i = tid; i < TWL; i += NT) twl[i] = tw[plan_tw_offset(LOG2H, 1) + i];
    cpx<double> v[PT];
#pragma unroll
    for (int i = 0; i < ITER0; ++i) {
        const int bf = tid + i * NT;
        if (FULL0 || bf < NB0) {
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int i0 = D * 2 * (bf + j * NB0) + ph, i1 = i0 + D;   // samples 2 m' and 2 m' + 1 of x_p
                v[i * R0 + j] = {i0 < n ? (double)x[i0] : 0.0, i1 < n ? (double)x[i1] : 0.0};
            }
        }
    }
    fft_all<double, LOG2H, false, PT, NT, false>(lds, tw, twl, tid, v);   // Z = FFT_H(x_p[2m] + i x_p[2m+1]) in lds[cpad(k)]
    // X_p(k) = E(k) + w_k O(k), E = (Z(k) + conj Z(H-k)) / 2, O = (Z(k) - conj Z(H-k)) / (2i); X_p(H-k) = conj(E - w_k O)
    cpx<float>* __restrict__ Xo = X + (size_t)ul * xpitch;
    cpx<double>* __restrict__ Xp = Xpart + (size_t)blockIdx.x * (H + 1);
    for (int k = 1 + tid; k <= H / 2; k += NT) {
        const cpx<double> zk = lds[cpad(k)], zh = lds[cpad(H - k)];
        const cpx<double> e = {0.5 * (zk.re + zh.re), 0.5 * (zk.im - zh.im)};
        const cpx<double> o = {0.5 * (zk.im + zh.im), -0.5 * (zk.re - zh.re)};
        const cpx<double> w = V[k];
        const cpx<double> wo = {o.re * w.re - o.im * w.im, o.re * w.im + o.im * w.re};
        if constexpr (LOGD == 0) {
            Xo[k] = {(float)(e.re + wo.re), (float)(e.im + wo.im)};
            Xo[H - k] = {(float)(e.re - wo.re), (float)(-(e.im - wo.im))};
        } else {
            Xp[k] = {e.re + wo.re, e.im + wo.im};
            Xp[H - k] = {e.re - wo.re, -(e.im - wo.im)};
        }
    }
    if (tid == 0) {
        const cpx<double> z0 = lds[cpad(0)];
        if constexpr (LOGD == 0) {
            Xo[0] = {(float)(z0.re + z0.im), 0.f};
            Xo[H] = {(float)(z0.re - z0.im), 0.f};
        } else {
            Xp[0] = {z0.re + z0.im, 0.0};
            Xp[H] = {z0.re - z0.im, 0.0};
        }
    }
}

// X(k) = sum_p W_M^(p k) X_p(k mod 2 HS), k = 0..M/2 = D HS; e64[q] = exp(-2 pi i q / M), q < M (float64)
template <int LOG2HS, int LOGD>
__global__ __launch_bounds__(256) void k_spectrum_combine(const cpx<double>* __restrict__ Xpart, const cpx<double>* __restrict__ e64,
                                                          cpx<float>* __restrict__ X, int64_t xpitch) {
    constexpr int D = 1 << LOGD, HS = 1 << LOG2HS, H = D * HS, M = 2 * H;
    const int ul = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > H) return;
    const int kk = k & (2 * HS - 1);
    double re = 0.0, im = 0.0;
#pragma unroll
    for (int ph = 0; ph < D; ++ph) {
        const cpx<double>* __restrict__ Xp = Xpart + ((size_t)ul * D + ph) * (HS + 1);
        cpx<double> xp = kk <= HS ? Xp[kk] : Xp[2 * HS - kk];
        if (kk > HS) xp.im = -xp.im;
        const cpx<double> w = e64[(ph * k) & (M - 1)];
        re += xp.re * w.re - xp.im * w.im;
        im += xp.re * w.im + xp.im * w.re;
    }
    X[(size_t)ul * xpitch + k] = {(float)re, (float)im};
}

// ---- state of the cascade at the end of every row -> digits of Q (float64), one wave = 64 channels of an utterance ----
constexpr int TAIL_TB = 32;
constexpr int TAIL_WAVES = 4;

template <typename WaveT>
__global__ __launch_bounds__((64 * TAIL_WAVES)) void k_tail_state(const WaveT* __restrict__ wave, const int64_t* __restrict__ offsets,
                                                                const double* __restrict__ coefs, int C, int groups,
                                                                const int* __restrict__ ulist, int nutt,
                                                                const int* __restrict__ Lgroup, float* __restrict__ rho,
                                                                double two_over_m) {
    __shared__ double xss[TAIL_WAVES][TAIL_TB];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * TAIL_WAVES + wid;
    if (unit >= nutt * groups) return;   // whole wave; the waves of a workgroup never meet at a barrier
    const int ul = unit / groups, g = unit - ul * groups;
    const int b = ulist[ul];
    const int64_t off = offsets[b];
    const int64_t N = offsets[b + 1] - off;
    const int cc = g * 64 + lane;
    const int c = min(cc, C - 1);
    const double* k = coefs + (size_t)c * 10;
    const double rB0 = 1.0 / k[6];
    const double b0 = k[0] * rB0, a1 = k[7] * rB0, a2 = k[8] * rB0;
    const double c1 = k[1] / k[0], c2 = k[2] / k[0], c3 = k[3] / k[0], c4 = k[4] / k[0];
    const double scale = (b0 * b0) * (b0 * b0) / k[9] * two_over_m;
    double* xs = xss[wid];
    const WaveT* w = wave + off;
    // the recurrences start from zero state TB * nblk samples before the end (samples before the utterance are zero)
    const int64_t L = min((int64_t)Lgroup[g], N);
    const int64_t nblk = (L + TAIL_TB - 1) / TAIL_TB;
    const int64_t t_begin = N - nblk * TAIL_TB;
    double z10 = 0, z11 = 0, z20 = 0, z21 = 0, z30 = 0, z31 = 0, z40 = 0, z41 = 0;
    auto sample = [&](int64_t t) { return (lane < TAIL_TB && t >= 0 && t < N) ? (double)w[t] : 0.0; };
    double xnext = sample(t_begin + lane);
    for (int64_t t0 = t_begin; t0 < N; t0 += TAIL_TB) {
        if (lane < TAIL_TB) xs[lane] = xnext;
        xnext = sample(t0 + TAIL_TB + lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the four sections skewed by one sample each (as K1): four independent float64 chains per step
        double p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
        for (int s2 = 0; s2 < TAIL_TB + 3; ++s2) {
            double n1 = 0, n2 = 0, n3 = 0;
            if (s2 < TAIL_TB) {
                const double wv = fma(-a1, z10, fma(-a2, z11, xs[s2]));
                n1 = fma(c1, z10, wv);
                z11 = z10;
                z10 = wv;
            }
            if (s2 >= 1 && s2 - 1 < TAIL_TB) {
                const double wv = fma(-a1, z20, fma(-a2, z21, p1));
                n2 = fma(c2, z20, wv);
                z21 = z20;
                z20 = wv;
            }
            if (s2 >= 2 && s2 - 2 < TAIL_TB) {
                const double wv = fma(-a1, z30, fma(-a2, z31, p2));
                n3 = fma(c3, z30, wv);
                z31 = z30;
                z30 = wv;
            }
            if (s2 >= 3) {
                const double wv = fma(-a1, z40, fma(-a2, z41, p3));
                z41 = z40;
                z40 = wv;
            }
            p1 = n1;
            p2 = n2;
            p3 = n3;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // residues R_m = (R_m0, R_m1) of the sections from their state words (w_m[n-1], w_m[n-2])
    const double ia2 = 1.0 / a2;
    const double R10 = (a1 - c1) * z10 + a2 * z11, R11 = a2 * z10 + c1 * a2 * z11;
    const double R20 = (a1 - c2) * z20 + a2 * z21, R21 = a2 * z20 + c2 * a2 * z21;
    const double R30 = (a1 - c3) * z30 + a2 * z31, R31 = a2 * z30 + c3 * a2 * z31;
    const double R40 = (a1 - c4) * z40 + a2 * z41, R41 = a2 * z40 + c4 * a2 * z41;
    double d0[2], d1[2], d2[2], d3[2];
    // R_4 D^3
    d3[0] = R40;
    d3[1] = R41;
    {   // N_4 R_3 D^2: degree 2 = q D + rem
        const double p0 = R30, p1q = R31 + c4 * R30, p2q = c4 * R31;
        const double q = p2q * ia2;
        d2[0] = p0 - q;
        d2[1] = p1q - q * a1;
        d3[0] += q;
    }
    {   // N_4 N_3 R_2 D: degree 3 = (q0 + q1 w) D + rem
        const double s0 = 1.0, s1 = c4 + c3, s2 = c4 * c3;   // N_4 N_3
        double p0 = s0 * R20, p1q = s0 * R21 + s1 * R20, p2q = s1 * R21 + s2 * R20, p3q = s2 * R21;
        const double q1 = p3q * ia2;
        p2q -= q1 * a1;
        p1q -= q1;
        const double q0 = p2q * ia2;
        p1q -= q0 * a1;
        p0 -= q0;
        d1[0] = p0;
        d1[1] = p1q;
        d2[0] += q0;
        d2[1] += q1;
    }
    {   // N_4 N_3 N_2 R_1: degree 4 = (q0 + q1 w + q2 w^2) D + rem; the quotient once more = qq D + rem'
        const double s0 = 1.0, s1 = c4 + c3, s2 = c4 * c3;
        const double t0 = s0, t1 = s1 + c2 * s0, t2 = s2 + c2 * s1, t3 = c2 * s2;   // N_4 N_3 N_2
        double p0 = t0 * R10, p1q = t0 * R11 + t1 * R10, p2q = t1 * R11 + t2 * R10, p3q = t2 * R11 + t3 * R10, p4q = t3 * R11;
        const double q2 = p4q * ia2;
        p3q -= q2 * a1;
        p2q -= q2;
        const double q1 = p3q * ia2;
        p2q -= q1 * a1;
        p1q -= q1;
        const double q0 = p2q * ia2;
        p1q -= q0 * a1;
        p0 -= q0;
        d0[0] = p0;
        d0[1] = p1q;
        const double qq = q2 * ia2;
        d1[0] += q0 - qq;
        d1[1] += q1 - qq * a1;
        d2[0] += qq;
    }
    if (cc < C) {
        float* r = rho + ((size_t)ul * C + cc) * 8;
        typedef float f4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f4*>(r) = f4{(float)(d0[0] * scale), (float)(d0[1] * scale), (float)(d1[0] * scale), (float)(d1[1] * scale)};
        *reinterpret_cast<f4*>(r + 4) = f4{(float)(d2[0] * scale), (float)(d2[1] * scale), (float)(d3[0] * scale), (float)(d3[1] * scale)};
    }
}

// samples after which n^3 r^n stays below tol x its peak (r = pole radius); 0 if the poles are not a complex pair
// inside the unit circle or the answer exceeds `limit`
// ---- the tables of a (coefficient table, length class), built on the device ----
// Row c < C: HU[c][q] = {Hs.re, Hs.im, u.re, u.im}, q = 0 .. H, u = 1 / D(w_q), Hs = (2 / M) s N_1 .. N_4 u^4 (float64 arithmetic,
// float32 out; entries H + 1 .. tpitch - 1 zero); row C: E[q] = w_q = exp(-2 pi i q / M), q < M (and its float64 copy for
// k_spectrum_combine). On eight host threads this took 29 / 51 / 93 ms of the first call of a length class (1 / 2 / 4 s rows,
// 128 channels): a one-shot `prepare envelope` over a few hundred files spent a third of its time there.
__global__ __launch_bounds__(256) void k_spectral_tables(const double* __restrict__ coefs, int C, int H, int64_t tpitch,
                                                         f2_f4* __restrict__ hu, cpx<float>* __restrict__ e,
                                                         cpx<double>* __restrict__ e64) {
    const int M = 2 * H;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    double sn, cs;
    sincospi(2.0 * (double)q / (double)M, &sn, &cs);
    const double wr = cs, wi = -sn;
    if (c == C) {
        if (q < M) {
            e[q] = {(float)wr, (float)wi};
            if (e64) e64[q] = {wr, wi};
        }
        return;
    }
    if (q >= tpitch) return;
    f2_f4 out = {0.f, 0.f, 0.f, 0.f};
    if (q <= H) {
        const double* k = coefs + (size_t)c * 10;
        const double rB0 = 1.0 / k[6];
        const double b0 = k[0] * rB0, a1 = k[7] * rB0, a2 = k[8] * rB0;
        const double s = (b0 * b0) * (b0 * b0) / k[9] * (2.0 / M);
        const double w2r = wr * wr - wi * wi, w2i = 2.0 * wr * wi;
        const double dr = 1.0 + a1 * wr + a2 * w2r, di = a1 * wi + a2 * w2i;
        const double dn = 1.0 / (dr * dr + di * di);
        const double ur = dr * dn, ui = -di * dn;
        double hr = s, hi = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const double cm = k[1 + m] / k[0];
            const double nr = 1.0 + cm * wr, ni = cm * wi;
            const double tr = nr * ur - ni * ui, ti = nr * ui + ni * ur;      // N_m u
            const double xr = hr * tr - hi * ti, xi = hr * ti + hi * tr;
            hr = xr;
            hi = xi;
        }
        out = {(float)hr, (float)hi, (float)ur, (float)ui};
    }
    hu[(size_t)c * tpitch + q] = out;
}

int64_t ringing_length(double a1, double a2, double tol, int64_t limit) {
    if (!(a2 > 1e-3 && a2 < 1.0) || a1 * a1 >= 4.0 * a2) return 0;
    const double lr = 0.5 * std::log(a2);   // log r < 0
    const double np = std::max(1.0, -3.0 / lr);
    const double target = std::log(tol) + 3.0 * std::log(np) + np * lr;   // log(tol * peak)
    double lo = np, hi = np;
    while (3.0 * std::log(hi) + hi * lr > target) {
        hi *= 2;
        if (hi > (double)limit) return 0;
    }
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (3.0 * std::log(mid) + mid * lr > target) lo = mid;
        else hi = mid;
    }
    return (int64_t)std::ceil(hi) + 64;
}

}  // namespace

// ---- host: eligibility, tables, launch ----
bool f2_spectral_supports_len(int64_t n, int min_pad) {
    if (n < 2) return false;
    const int log2h = f2_log2_ceil(n) - 1;
    if (log2h < F2_SPECTRAL_MIN_LOG2H || log2h > F2_SPECTRAL_MAX_LOG2H) return false;
    // The accuracy guard reads the error off the padding samples [n, M). The part of it that is coherent across the bins -
    // the float32 rounding of the ringing term's digits - is itself ringing-shaped: it starts at n, peaks 3 / -ln(r) samples
    // later and wraps round into the START of the row. A row with less padding than that would hide the peak from the guard
    // (tests/diag/guard_search.py: 1.3e-5 unflagged at 64 padding samples), so it keeps the two-kernel route.
    return (int64_t(2) << log2h) - n >= std::max(64, min_pad);
}

bool f2_spectral_supports_coefs(const std::vector<double>& coefs, int C, std::vector<int>* Lgroup, int* min_pad) {
    if (coefs.size() != (size_t)C * 10 || C <= 0) return false;
    const int groups = (C + 63) / 64;
    if (Lgroup) Lgroup->assign((size_t)groups, 0);
    double peak = 0.0;
    for (int c = 0; c < C; ++c) {
        const double* k = &coefs[(size_t)c * 10];
        if (k[5] != 0.0 || k[0] == 0.0 || k[6] == 0.0 || k[9] == 0.0) return false;
        for (int q = 0; q < 10; ++q)
            if (!std::isfinite(k[q])) return false;
        const int64_t L = ringing_length(k[7] / k[6], k[8] / k[6], 1e-8, int64_t(1) << 17);
        if (L <= 0) return false;
        if (Lgroup) (*Lgroup)[(size_t)c / 64] = std::max((*Lgroup)[(size_t)c / 64], (int)L);
        peak = std::max(peak, 3.0 / (-0.5 * std::log(k[8] / k[6])));   // n^3 r^n peaks at n = 3 / -ln r
    }
    if (min_pad) *min_pad = ((int)std::ceil(1.1 * peak) + 31) / 32 * 32;   // 256 for the 100 Hz channel of the reference's bank
    return true;
}

static int spectral_tables(f2_ctx* ctx, const double* d_coefs, int C, int log2h, f2_spec_tables** out) {
    for (auto& t : ctx->spec_tabs)
        if (t.log2h == log2h && t.C == C && t.coefs == ctx->coefs_host) {
            *out = &t;
            return F2_OK;
        }
    if (ctx->spec_tabs.size() >= 6) {   // a handful of (table, length class) pairs at most; drop the oldest
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (f2_scratch* s : {&ctx->spec_tabs.front().hu, &ctx->spec_tabs.front().e, &ctx->spec_tabs.front().lgroup,
                              &ctx->spec_tabs.front().e64})
            if (s->ptr) (void)hipFree(s->ptr);
        ctx->spec_tabs.erase(ctx->spec_tabs.begin());
    }
    f2_spec_tables t;
    t.log2h = log2h;
    t.C = C;
    t.coefs = ctx->coefs_host;
    std::vector<int> Lg;
    if (!f2_spectral_supports_coefs(t.coefs, C, &Lg, nullptr)) return f2_fail(ctx, F2_ERR_INVALID, "coefficient table not eligible");
    const int H = 1 << log2h, M = 2 * H;
    t.tpitch = H + 8;
    F2_TRY(f2_reserve(ctx, t.hu, sizeof(float) * 4 * (size_t)C * (size_t)t.tpitch));
    F2_TRY(f2_reserve(ctx, t.e, sizeof(float) * 2 * (size_t)M));
    F2_TRY(f2_reserve(ctx, t.lgroup, sizeof(int) * Lg.size()));
    if (log2h > 13) F2_TRY(f2_reserve(ctx, t.e64, sizeof(double) * 2 * (size_t)M));   // float64 copy for k_spectrum_combine (decimated utterances)
    hipLaunchKernelGGL(k_spectral_tables, dim3((unsigned)((M + 255) / 256), (unsigned)(C + 1)), dim3(256), 0, ctx->stream, d_coefs, C, H,
                       t.tpitch, (f2_f4*)t.hu.ptr, (cpx<float>*)t.e.ptr, (cpx<double>*)t.e64.ptr);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_upload_async(ctx, t.lgroup.ptr, Lg.data(), sizeof(int) * Lg.size()));
    ctx->spec_tabs.push_back(std::move(t));
    *out = &ctx->spec_tabs.back();
    return F2_OK;
}

// powers of q = -a1 every thread needs (lowpass_pairs_store_tab: sp = 2 samples per thread and block), cached per a1
static int lowpass_table(f2_ctx* ctx, double a1, double b0, int nt, int sp, LowpassConsts* K, const f2_f4** d_tab) {
    const long double q = -(long double)a1;
    K->qf = (float)q;
    K->b0f = (float)b0;
    K->g1 = (float)powl(q, sp);
    K->g2 = (float)powl(q, 2 * sp);
    K->g4 = (float)powl(q, 4 * sp);
    K->g8 = (float)powl(q, 8 * sp);
    K->gw = (double)powl(q, 64 * sp);
    K->gblk = (double)powl(q, sp * nt);
    f2_scratch& slot = ctx->spec_lptab;
    double& slot_a1 = ctx->spec_lptab_a1;
    *d_tab = (const f2_f4*)slot.ptr;
    constexpr int TMAX = 1024;   // the per-thread entries do not depend on the workgroup size: one table serves all
    if (slot.ptr && slot_a1 == a1) return F2_OK;
    std::vector<float> tab((size_t)TMAX * 4);
    for (int t = 0; t < TMAX; ++t) {
        const int lane = t & 63;
        tab[4 * (size_t)t] = (float)powl(q, sp * (lane + 1));
        tab[4 * (size_t)t + 1] = (float)powl(q, sp * t);
        tab[4 * (size_t)t + 2] = (float)powl(q, sp * ((lane & 15) + 1));
        tab[4 * (size_t)t + 3] = (float)powl(q, sp * ((lane & 31) + 1));
    }
    F2_TRY(f2_reserve(ctx, slot, sizeof(float) * tab.size()));
    F2_TRY(f2_upload_async(ctx, slot.ptr, tab.data(), sizeof(float) * tab.size()));
    slot_a1 = a1;
    *d_tab = (const f2_f4*)slot.ptr;
    return F2_OK;
}

template <typename WaveT, int LOG2H>
static int launch_group(f2_ctx* ctx, const WaveT* d_wave, const int64_t* d_offsets, const double* d_coefs, int C,
                        const int* d_ulist, int nutt, int64_t min_n, int lpf, double b0, double a1, double* d_env, int* d_uflag,
                        float tol, cpx<float>* d_X, float* d_rho) {
    constexpr int H = 1 << LOG2H, M = 2 * H;
    f2_spec_tables* tab = nullptr;
    F2_TRY(spectral_tables(ctx, d_coefs, C, LOG2H, &tab));
    // float64 transform of the utterances: 2^13 packed complex points fit in LDS; longer rows are decimated in time
    constexpr int LOGD = LOG2H > 13 ? LOG2H - 13 : 0, LOG2HS = LOG2H - LOGD;
    constexpr int FFTLOG = LOG2H == 15 ? 14 : LOG2H;   // rows of the longest class: four 16384-point transforms each
    F2_TRY(ensure_twiddles<double>(ctx, LOG2HS, ctx->tw_sp[1][LOG2HS]));
    F2_TRY(ensure_twiddles<float>(ctx, FFTLOG, ctx->tw_sp[0][FFTLOG]));
    const int64_t xpitch = H + 8;
    const int groups = (C + 63) / 64;
    F2_TRY(f2_prof_begin(ctx, F2_K_SPECTRUM));
    cpx<double>* d_part = nullptr;
    if constexpr (LOGD > 0) {
        F2_TRY(f2_reserve(ctx, ctx->spec_xpart, sizeof(double) * 2 * ((size_t)(1 << LOG2HS) + 1) * (size_t)nutt << LOGD));
        d_part = (cpx<double>*)ctx->spec_xpart.ptr;
    }
    hipLaunchKernelGGL((k_utterance_spectrum<WaveT, LOG2HS, LOGD>), dim3((unsigned)nutt << LOGD), dim3(threads_for<double, LOG2HS>()),
                       0, ctx->stream, d_wave, d_offsets, d_ulist, d_X, xpitch, d_part,
                       (const cpx<double>*)ctx->tw_sp[1][LOG2HS].ptr);
    F2_HIP(ctx, hipGetLastError());
    if constexpr (LOGD > 0) {
        hipLaunchKernelGGL((k_spectrum_combine<LOG2HS, LOGD>), dim3((unsigned)((H + 256) / 256), (unsigned)nutt), dim3(256), 0,
                           ctx->stream, (const cpx<double>*)d_part, (const cpx<double>*)tab->e64.ptr, d_X, xpitch);
        F2_HIP(ctx, hipGetLastError());
    }
    F2_TRY(f2_prof_end(ctx, F2_K_SPECTRUM));
    F2_TRY(f2_prof_begin(ctx, F2_K_TAIL));
    const int units = nutt * groups;
    hipLaunchKernelGGL((k_tail_state<WaveT>), dim3((unsigned)((units + TAIL_WAVES - 1) / TAIL_WAVES)), dim3(64 * TAIL_WAVES), 0,
                       ctx->stream, d_wave, d_offsets, d_coefs, C, groups, d_ulist, nutt, (const int*)tab->lgroup.ptr, d_rho,
                       2.0 / M);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_TAIL));
    SpecParams P;
    P.xpitch = xpitch;
    P.tpitch = tab->tpitch;
    P.env = d_env;
    P.uflag = d_uflag;
    P.C = C;
    P.nutt = nutt;
    P.lpf = lpf;
    const f2_f4* d_lptab = nullptr;
    F2_TRY(lowpass_table(ctx, a1, b0, threads_for<float, FFTLOG>(), 2, &P.lp, &d_lptab));
    P.tol = tol;
    P.gdump = ctx->opt_spectral_guard_dump ? (float*)ctx->spec_gdump.ptr : nullptr;
    P.stamps = nullptr;
#ifdef F2_STAMPS
    static unsigned long long* d_stamps = nullptr;
    const size_t nstamp = (size_t)nutt * C * 8;
    if (!d_stamps) F2_HIP(ctx, hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 8 * 128 * 2048));
    if (nstamp <= (size_t)8 * 128 * 2048) P.stamps = d_stamps;
#endif
    F2_TRY(f2_prof_begin(ctx, F2_K_FUSED));
    if constexpr (LOG2H == 15)
        hipLaunchKernelGGL(k_spectral_envelope_long, dim3((unsigned)((size_t)nutt * C)), dim3(threads_for<float, FFTLOG>()), 0,
                           ctx->stream, P, (const cpx<float>*)d_X, (const f2_f4*)tab->hu.ptr, (const cpx<float>*)tab->e.ptr,
                           (const float*)d_rho, d_offsets, d_ulist, d_lptab, (const cpx<float>*)ctx->tw_sp[0][FFTLOG].ptr);
    else if (min_n >= 15 * (M / 16))    // every row's padding lies in the last block: the guard looks at that block only
        hipLaunchKernelGGL((k_spectral_envelope<FFTLOG, true>), dim3((unsigned)((size_t)nutt * C)), dim3(threads_for<float, FFTLOG>()), 0,
                           ctx->stream, P, (const cpx<float>*)d_X, (const f2_f4*)tab->hu.ptr, (const cpx<float>*)tab->e.ptr,
                           (const float*)d_rho, d_offsets, d_ulist, d_lptab, (const cpx<float>*)ctx->tw_sp[0][FFTLOG].ptr);
    else
        hipLaunchKernelGGL((k_spectral_envelope<FFTLOG, false>), dim3((unsigned)((size_t)nutt * C)), dim3(threads_for<float, FFTLOG>()), 0,
                           ctx->stream, P, (const cpx<float>*)d_X, (const f2_f4*)tab->hu.ptr, (const cpx<float>*)tab->e.ptr,
                           (const float*)d_rho, d_offsets, d_ulist, d_lptab, (const cpx<float>*)ctx->tw_sp[0][FFTLOG].ptr);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_FUSED));
#ifdef F2_STAMPS
    if (P.stamps) {
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<unsigned long long> h(nstamp);
        F2_HIP(ctx, hipMemcpy(h.data(), d_stamps, sizeof(unsigned long long) * nstamp, hipMemcpyDeviceToHost));
        double acc[8] = {0};
        const size_t rows = (size_t)nutt * C;
        unsigned long long t0 = ~0ull, t1 = 0;
        double life = 0;
        for (size_t r = 0; r < rows; ++r) {
            for (int k = 1; k < 8; ++k) acc[k] += (double)(h[r * 8 + k] - h[r * 8 + k - 1]);
            t0 = std::min(t0, h[r * 8]);
            t1 = std::max(t1, h[r * 8 + 7]);
            life += (double)(h[r * 8 + 7] - h[r * 8]);
        }
        static const char* names_s[8] = {"", "bin 0 ready", "bin 7 ready", "bin 15 ready", "transform inputs", "both transforms + even magnitudes", "odd magnitudes + guard", "lpf+stores"};
        static const char* names_l[8] = {"", "spectrum r=0,2", "transforms r=0,2 + parking", "spectrum r=1,3", "transforms r=1,3 + parking", "low-pass sweep 0", "low-pass sweep 1", "flag"};
        const char* const* names = LOG2H == 15 ? names_l : names_s;
        fprintf(stderr, "[stamps KS] mean ticks (10 ns) per workgroup:");
        for (int k = 1; k < 8; ++k) fprintf(stderr, " %s=%.0f", names[k], acc[k] / rows);
        fprintf(stderr, "\n[stamps KS] mean workgroup lifetime %.1f ticks, kernel span %.0f ticks, workgroups alive at once %.1f\n",
                life / rows, (double)(t1 - t0), life / (double)(t1 - t0));
    }
#endif
    return F2_OK;
}

// Envelopes of the utterances `utts` (host list, all of length class log2h and eligible) straight from the waves.
int f2_launch_spectral(f2_ctx* ctx, const void* d_wave, int wave_dtype, const int64_t* d_offsets, const double* d_coefs,
                       int C, const int* d_ulist, int nutt, int64_t min_n, int log2h, int lpf, double cutoff_hz, double* d_env,
                       int* d_uflag) {
    if (nutt <= 0) return F2_OK;
    const double k = lpf ? tan(3.14159265358979323846 * cutoff_hz / 16000.0) : 0.0;
    const double b0 = k / (1.0 + k), a1 = (k - 1.0) / (k + 1.0);
    const float tol = ctx->opt_spectral_tol;
    const int H = 1 << log2h;
    F2_TRY(f2_reserve(ctx, ctx->spec_x, sizeof(float) * 2 * (size_t)(H + 8) * (size_t)nutt));
    F2_TRY(f2_reserve(ctx, ctx->spec_rho, sizeof(float) * 8 * (size_t)nutt * (size_t)C));
    cpx<float>* d_X = (cpx<float>*)ctx->spec_x.ptr;
    float* d_rho = (float*)ctx->spec_rho.ptr;
#define F2_SPEC_CASE(L)                                                                                                          \
    case L:                                                                                                                      \
        return wave_dtype == F2_WAVE_I16                                                                                         \
                   ? launch_group<int16_t, L>(ctx, (const int16_t*)d_wave, d_offsets, d_coefs, C, d_ulist, nutt, min_n, lpf, b0, \
                                              a1, d_env, d_uflag, tol, d_X, d_rho)                                              \
                   : launch_group<double, L>(ctx, (const double*)d_wave, d_offsets, d_coefs, C, d_ulist, nutt, min_n, lpf, b0,   \
                                             a1, d_env, d_uflag, tol, d_X, d_rho);
    switch (log2h) {
        F2_SPEC_CASE(12)
        F2_SPEC_CASE(13)
        F2_SPEC_CASE(14)
        F2_SPEC_CASE(15)
        default:
            return f2_fail(ctx, F2_ERR_UNSUPPORTED, "spectral path: length class 2^%d not built", log2h + 1);
    }
#undef F2_SPEC_CASE
}
