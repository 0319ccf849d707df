// K2 -- Hilbert-magnitude envelope + optional first-order Butterworth low-pass
// (reference: scripts/processing/EnvelopeExtraction.py:20-67 paddedHilbert / lowPassFilter /
// ExtractEnvelopeFromMatrix).
//
// One workgroup (256, 512 or 1024 threads: 16 points per thread for the benchmark sizes) owns one (utterance,
// channel) row of n samples, zero-padded to M = 2^ceil(log2 n) exactly as the reference does, and keeps the whole
// transform on chip:
//
//   1. the real row is read as H = M/2 complex points  z[m] = x[2m] + i x[2m+1]  (16-byte loads of float64 rows, or
//      8-byte loads of the filterbank kernel's float32 hand-off) straight into the registers of the first FFT pass;
//      float transforms keep their x values in registers for step 5
//   2. Z = FFT_H(z): Stockham passes with a symmetric radix plan (first radix = last radix; 16-8-4-16 for H = 8192,
//      f2_fft_lds.h); data moves registers -> LDS -> registers -> ..., every pass holds its inputs in registers across
//      the barrier so the transform is in place. Twiddles of passes >= 1 are copied to LDS once per workgroup; a
//      radix-16 first pass derives its 15 per-butterfly twiddles from two loaded ones
//   3. the Hilbert transform h = H[x] is real, so its packed spectrum follows from Z by one pass over the
//      pairs (k, H-k):  W[k] = i sin(t_k) Z[k] + cos(t_k) conj(Z[H-k]),  t_k = 2 pi k / M,  W[0] = 0
//      (derivation in DESIGN.md); scipy.signal.hilbert's analytic signal is x + i h
//   4. w = IFFT_H(W), run as conj(FFT(conj W)) (only squares of w are used); because first and last radix
//      are equal, the last pass leaves in each thread exactly the points it loaded in step 1:
//      h[2m] = Re w[m], h[2m+1] = Im w[m]
//   5. env[n] = sqrt(x[n]^2 + h[n]^2) from registers; without low-pass it is stored at once (16-byte stores). With
//      low-pass, y[n] = b0 (env[n] + env[n-1]) - a1 y[n-1] runs time-parallel in the register layout the FFT left
//      (lowpass_pairs_store, f2_envelope_core.h: pair -> DPP wave scan -> wave totals -> block chain) and the
//      outputs leave as 16-byte stores; rows whose first pass does not give every thread whole butterflies (small H)
//      go through LDS instead: contiguous chunk per thread, zero-state run, multiplicative scan of the chunk ends,
//      second run, with the envelope TRANSPOSED in LDS (sample t*L+j at j*(NT+1)+t) so sweeps and copy-out are
//      conflict free.
//
// Two real FFTs of length M cost two complex FFTs of length M/2 on 8-byte (f32) points: 64 KiB (+pad) of
// LDS for the 1 s / 16 kHz row of the benchmark, two workgroups per CU.
// Bound: HBM, 12-16 bytes per sample-channel (4 or 8 read + 8 written). Rows that do not fit in LDS: f2_envelope_split.hip
// (four-step transform), f2_envelope_large.hip (global-memory passes).
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "f2_fft_lds.h"

using namespace f2fft;

namespace {

// Diagnostic build only (-DF2_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at the phase boundaries.
#ifdef F2_STAMPS
#define F2_STAMP(k)                                          \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        st[k] = __builtin_amdgcn_s_memrealtime();              \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#else
#define F2_STAMP(k) \
    do {            \
    } while (0)
#endif


// (f2_envelope_flagged.hip compiles this file once more with F2_ENVELOPE_FLAGGED_TU: the body below then becomes the
// device function envelope_row(P, tw, u, c) that its looping kernel calls; this translation unit's kernel is unchanged)
template <typename F, int LOG2H>
#ifdef F2_ENVELOPE_FLAGGED_TU
__device__ __forceinline__ void envelope_row(const EnvParams& P, const cpx<F>* __restrict__ tw, const int u, const int c) {
#else
__global__ __launch_bounds__((threads_for<F, LOG2H>()), (min_waves_for<F, LOG2H>())) void k_envelope(EnvParams P, const cpx<F>* __restrict__ tw) {
#endif
    constexpr int NT = threads_for<F, LOG2H>();
    constexpr int H = 1 << LOG2H;
    constexpr int M = 2 * H;
    constexpr int CS = cpad_size(H);
    constexpr int L = M >= NT ? M / NT : 1;   // samples per thread in the low-pass (contiguous chunk)
    constexpr int TP = NT + 1;                // row pitch of the transposed envelope image
    constexpr int RS = L * TP;
    constexpr int LDS_BYTES = (CS * 2 > RS ? CS * 2 : RS) * (int)sizeof(F);
    // register shape of the first/last pass
    constexpr int R0 = 1 << plan_bits(LOG2H, 0);
    constexpr int NB0 = H / R0;
    constexpr int ITER0 = (NB0 + NT - 1) / NT;
    constexpr int PT = plan_points_per_thread(LOG2H, NT);   // complex points per thread (all passes)
    constexpr bool KEEP_X = sizeof(F) == 4 && 2 * ITER0 * R0 <= 64;   // x stays in registers for step 5
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    __shared__ double wave_tot[NT / 64];
    __shared__ F qpow[L];   // (-a1)^(j+1), j < L
    constexpr int TWL = plan_tw_lds_count(LOG2H);
    __shared__ __attribute__((aligned(16))) cpx<F> twl[TWL > 0 ? TWL : 1];   // twiddles of passes >= 1
    cpx<F>* lds = reinterpret_cast<cpx<F>*>(smem);
    F* rl = reinterpret_cast<F*>(smem);
    const cpx<F>* __restrict__ V = tw + plan_tw_total(LOG2H);   // exp(-2 pi i k / M), k <= H/2

    const int tid = threadIdx.x;
#ifndef F2_ENVELOPE_FLAGGED_TU
    const int u = blockIdx.x / P.C;
    const int c = blockIdx.x - u * P.C;
#endif
    const int b = P.ulist ? P.ulist[u] : u;
    if (P.uflag && !P.uflag[b]) return;   // (whole workgroup) this utterance was served by the spectral kernel
    const int64_t off = P.offsets[b];
    const int n = (int)(P.offsets[b + 1] - off);
    const size_t row = (size_t)P.C * (size_t)off + (size_t)c * (size_t)n;
#ifdef F2_KO_LOAD   // knock-out (timing only): every row reads the first row, so the input comes from cache
    const double* __restrict__ x = P.gfb;
#else
    const double* __restrict__ x = P.gfb + row;
#endif
    double* __restrict__ y = P.env + row;

#ifdef F2_STAMPS
    unsigned long long st[10] = {0};
#endif
    F2_STAMP(0);
    // (visible to every wave after the first pass's barrier; pass 0 never reads it)
    for (int i = tid; i < TWL; i += NT) twl[i] = tw[plan_tw_offset(LOG2H, 1) + i];
    // 1. load: point (i, j) of this thread is m = tid + i*NT + j*NB0
    constexpr bool FULL0 = NB0 % NT == 0;
    cpx<F> v[PT];
    [[maybe_unused]] F xr[KEEP_X ? ITER0 * R0 : 1], xi[KEEP_X ? ITER0 * R0 : 1];
    // branch-free inside each variant (clamped address + select) so that all loads are in flight together
    auto load_row = [&](auto* __restrict__ xs, auto odd) {
#pragma unroll
        for (int i = 0; i < ITER0; ++i) {
            const int bf = tid + i * NT;
            if (FULL0 || bf < NB0) {
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    F a, bb;
                    row_pair<decltype(odd)::value>(xs, n, 2 * (bf + j * NB0), a, bb);
                    v[i * R0 + j] = {a, bb};
                    if constexpr (KEEP_X) {
                        xr[i * R0 + j] = a;
                        xi[i * R0 + j] = bb;
                    }
                }
            }
        }
    };
    if (n < 2) {
        // a one-sample row: M = 1, no pair to load
        const F a = (KEEP_X && P.f32_in) ? (F) reinterpret_cast<const float*>(x)[0] : (F)x[0];
#pragma unroll
        for (int q = 0; q < ITER0 * R0; ++q) {
            v[q] = {F(0), F(0)};
            if constexpr (KEEP_X) xr[q] = xi[q] = F(0);
        }
        if (tid == 0) {
            v[0] = {a, F(0)};
            if constexpr (KEEP_X) xr[0] = a;
        }
    } else if (KEEP_X && P.f32_in) {
        // float32 hand-off from the filterbank kernel: the row's samples sit at the start of its float64 slot
        const float* __restrict__ xf = reinterpret_cast<const float*>(x);
        if ((n & 1) == 0) load_row(xf, std::false_type{});
        else load_row(xf, std::true_type{});
    } else {
        if ((n & 1) == 0) load_row(x, std::false_type{});
        else load_row(x, std::true_type{});
    }
    F2_STAMP(1);
#ifdef F2_FUSE_PROBE
    // Timing probe (results wrong by ~1e-30): the float64 work a row-parallel filterbank inside this workgroup would
    // add - F2_FUSE_PROBE FMAs per thread in eight independent chains - to see how it co-schedules with the
    // transforms of the other workgroup on the CU. Use with -DF2_KO_LOAD.
    {
        double acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = (double)v[q].re;
        const double ca = 0.999999 + 1e-9 * tid, cb = 1e-7;
        for (int it = 0; it < F2_FUSE_PROBE / 8; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fma(acc[q], ca, cb);
        }
        double sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += acc[q];
        v[0].re += (F)(sum * 1e-30);
    }
#endif
    // 2. forward transform (first pass straight from the registers)
    constexpr bool T0R = derive_tw0<F, LOG2H>();
    fft_all<F, LOG2H, false, PT, NT, T0R>(lds, tw, twl, tid, v);
    F2_STAMP(2);
    // 3. packed spectrum of the Hilbert transform, conjugated and scaled by 1/H for step 4:
    //    W[k] = i sin(t_k) Z[k] + cos(t_k) conj(Z[H-k]), W[0] = 0. Either folded into the first pass of
    //    step 4 (fuse_hilbert) or one sweep over the pairs (k, H-k) here, all loads issued up front.
    if constexpr (LOG2H >= 1 && !fuse_hilbert<F, LOG2H>()) {
        constexpr int NPAIR = H / 2 - 1;                       // pairs (k, H-k), 1 <= k < H/2
        constexpr int ITERP = (NPAIR + NT - 1) / NT;
        const F sc = F(1.0 / H);
        if constexpr (NPAIR > 0) {
            cpx<F> zk[ITERP], zh[ITERP], vk[ITERP];
#pragma unroll
            for (int i = 0; i < ITERP; ++i) {
                const int k = min(1 + tid + i * NT, H / 2 - 1);
                zk[i] = lds[cpad(k)];
                zh[i] = lds[cpad(H - k)];
                vk[i] = V[k];                                   // (cos t, -sin t)
            }
#pragma unroll
            for (int i = 0; i < ITERP; ++i) {
                const int k = 1 + tid + i * NT;
                const F cs = vk[i].re * sc, sn = -vk[i].im * sc;
                // W[k] = i sn Z[k] + cs conj(Z[H-k]);  W[H-k] = i sn Z[H-k] - cs conj(Z[k]); stored conjugated
                if (k < H / 2) {
                    lds[cpad(k)] = {-sn * zk[i].im + cs * zh[i].re, -(sn * zk[i].re - cs * zh[i].im)};
                    lds[cpad(H - k)] = {-sn * zh[i].im - cs * zk[i].re, -(sn * zh[i].re + cs * zk[i].im)};
                }
            }
        }
        if (tid == 0) {
            const cpx<F> zq = lds[cpad(H / 2)];                 // t = pi/2: W = i Z -> conj(W) = (-zim, -zre)
            lds[cpad(H / 2)] = {-zq.im * sc, -zq.re * sc};
            lds[cpad(0)] = {F(0), F(0)};
        }
        __syncthreads();
    }
    F2_STAMP(3);
    // 4. inverse transform (forward transform of the conjugate); outputs stay in v
    fft_all<F, LOG2H, true, PT, NT, T0R>(lds, tw, twl, tid, v);
    if constexpr (LOG2H == 0) v[0] = {F(0), F(0)};   // H = 1: W[0] = 0, the envelope is |x|

    F2_STAMP(4);
    // 5. magnitude (in F: the FFT already limits the accuracy to F). The last pass left point j of
    //    butterfly i in v[i*R0 + brev(j)]; the envelope pair replaces it there.
#pragma unroll
    for (int i = 0; i < ITER0; ++i) {
        const int bf = tid + i * NT;
        if (FULL0 || bf < NB0) {
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int i0 = 2 * (bf + j * NB0);
                F a, bb;
                if constexpr (KEEP_X) {
                    a = xr[i * R0 + j];
                    bb = xi[i * R0 + j];
                } else {
                    a = i0 < n ? (F)x[i0] : F(0);
                    bb = i0 + 1 < n ? (F)x[i0 + 1] : F(0);
                }
                const cpx<F> w = v[i * R0 + brev<R0>(j)];
                v[i * R0 + brev<R0>(j)] = {fsqrt(a * a + w.re * w.re), fsqrt(bb * bb + w.im * w.im)};
            }
        }
    }
    if (!P.lpf) {
#pragma unroll
        for (int i = 0; i < ITER0; ++i) {
            const int bf = tid + i * NT;
            if (FULL0 || bf < NB0) {
#pragma unroll
                for (int j = 0; j < R0; ++j) {
                    const cpx<F> e = v[i * R0 + brev<R0>(j)];
                    // (the points of a butterfly index j are NB0 apart: block j of 2 NB0 samples; inside the row - a wave-uniform
                    // test - it stores without a test per lane)
                    if (FULL0 && 2 * NB0 * (j + 1) <= n) store_pair(y + 2 * (bf + j * NB0), (double)e.re, (double)e.im);
                    else store_row_pair(y, n, 2 * (bf + j * NB0), (double)e.re, (double)e.im);
                }
            }
        }
        return;
    }
    F2_STAMP(5);
    if constexpr (FULL0) {
        // block jj = i + ITER0*j of this thread's envelope pairs sits in v[i*R0 + brev(j)]
        constexpr int NBLK = ITER0 * R0;
        static_assert(lowpass_lds_bytes<F, NT, NBLK>() <= (size_t)LDS_BYTES, "LDS too small");
        F er[NBLK], ei[NBLK];
#pragma unroll
        for (int i = 0; i < ITER0; ++i)
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                er[i + ITER0 * j] = v[i * R0 + brev<R0>(j)].re;
                ei[i + ITER0 * j] = v[i * R0 + brev<R0>(j)].im;
            }
        lowpass_pairs_store<F, NT, NBLK>(er, ei, P.a1, P.b0, smem, y, n, tid);
#ifdef F2_STAMPS
        F2_STAMP(6);
        st[7] = st[8] = st[9] = st[6];
        if (tid == 0 && P.stamps)
            for (int k = 0; k < 10; ++k) P.stamps[(size_t)blockIdx.x * 10 + k] = st[k];
#endif
        return;
    }
    // With the low-pass the envelope goes to LDS, TRANSPOSED: thread t will own the contiguous samples
    // [t*L, (t+1)*L), so sample t*L + j is kept at j*TP + t. (The last FFT pass ended with a barrier after
    // its LDS reads, so the array is free.)
    auto tpos = [](int i) { return (i % L) * TP + i / L; };
#pragma unroll
    for (int i = 0; i < ITER0; ++i) {
        const int bf = tid + i * NT;
        if (FULL0 || bf < NB0) {
#pragma unroll
            for (int j = 0; j < R0; ++j) {
                const int i0 = 2 * (bf + j * NB0);
                rl[tpos(i0)] = v[i * R0 + brev<R0>(j)].re;
                rl[tpos(i0 + 1)] = v[i * R0 + brev<R0>(j)].im;
            }
        }
    }
    __syncthreads();

    F2_STAMP(6);
    // low-pass: chunked recurrence + multiplicative scan over the chunks. Each thread runs its chunk once
    // from zero state (float64 accumulator, float32 input term) keeping the zero-state responses; the true
    // output is that plus carry * (-a1)^(j+1), one float FMA per sample with the powers from an LDS table.
    const int n0 = tid * L;
    const double na1 = -P.a1;
    const F b0f = (F)P.b0;
    const F eprev = (n0 > 0 && n0 <= n) ? rl[tpos(n0 - 1)] : F(0);
    if (tid < L) {
        double pw = na1;
        for (int j = 0; j < tid; ++j) pw *= na1;
        qpow[tid] = (F)pw;   // (-a1)^(tid+1)
    }
    // The chunk is cut into NS sub-chunks whose recurrences run interleaved (independent float64 chains:
    // the FMA latency is ~25 cycles, a single 32-sample chain would cost more than the arithmetic).
    constexpr int NS = L >= 64 ? 8 : L >= 16 ? 4 : 1;   // sub-chunks of at most 16 samples
    constexpr int LS = L / NS;
    F yzs[L];                 // zero-state response of every sub-chunk
    double zend[NS];
    {
        // inside a sub-chunk (<= 16 samples) the recurrence runs in F: the rounding of so few steps stays at a few
        // ulp of the (positive) partial sums; everything that crosses sub-chunks, chunks and waves is float64
        F ep[NS], z[NS];
        const F na1f = (F)na1;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            z[q] = F(0);
            ep[q] = q == 0 ? eprev : rl[(q * LS - 1) * TP + tid];
        }
#pragma unroll
        for (int j = 0; j < LS; ++j) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const F e = rl[(q * LS + j) * TP + tid];   // samples past n hold |0| = 0
                z[q] = na1f * z[q] + b0f * (e + ep[q]);
                yzs[q * LS + j] = z[q];
                ep[q] = e;
            }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) zend[q] = (double)z[q];
    }
    // gs = (-a1)^LS, g = (-a1)^L
    double gs = na1;
#pragma unroll
    for (int s2 = 1; s2 < LS; s2 <<= 1) gs *= gs;
    double g = gs;
#pragma unroll
    for (int s2 = 1; s2 < NS; s2 <<= 1) g *= g;
    // chunk-level zero-state value at the end of every sub-chunk
    double ysub[NS];
    ysub[0] = zend[0];
#pragma unroll
    for (int q = 1; q < NS; ++q) ysub[q] = fma(gs, ysub[q - 1], zend[q]);
    const double yz = ysub[NS - 1];
    // inclusive scan inside the wave: v_t = sum_{j<=t} g^(t-j) yz_j
    const int lane = tid & 63, wv = tid >> 6;
    double sc = yz, gd = g;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = shfl_up_f64(sc, d);
        if (lane >= d) sc = fma(gd, up, sc);
        gd *= gd;
    }
    // gd == g^64 now
    F2_STAMP(7);
    if (lane == 63) wave_tot[wv] = sc;
    // g^(lane+1) (data independent; placed before the barrier to overlap with the other waves' scans)
    double gl = 1.0, gp = g;
    for (int bits = lane + 1; bits; bits >>= 1) {
        if (bits & 1) gl *= gp;
        gp *= gp;
    }
    __syncthreads();
    double carry = 0.0;  // true y at the end of the previous wave's last chunk
    for (int w2 = 0; w2 < wv; ++w2) carry = fma(gd, carry, wave_tot[w2]);
    const double incl = fma(gl, carry, sc);                 // true y at the end of this chunk
    double yprev = shfl_up_f64(incl, 1);                    // ... of the previous chunk
    if (lane == 0) yprev = carry;
    {
        // value entering sub-chunk q: chunk-level zero-state part + (-a1)^(q*LS) * yprev
        double gq = 1.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const F cq = (F)(q == 0 ? yprev : fma(gq, yprev, ysub[q - 1]));
#pragma unroll
            for (int j = 0; j < LS; ++j) rl[(q * LS + j) * TP + tid] = cq * qpow[j] + yzs[q * LS + j];
            gq *= gs;
        }
    }
    __syncthreads();
    F2_STAMP(8);
#pragma unroll 4
    for (int i = 2 * tid; i < n; i += 2 * NT) store_row_pair(y, n, i, (double)rl[tpos(i)], (double)rl[tpos(min(i + 1, n - 1))]);
#ifdef F2_STAMPS
    F2_STAMP(9);
    if (tid == 0 && P.stamps)
        for (int k = 0; k < 10; ++k) P.stamps[(size_t)blockIdx.x * 10 + k] = st[k];
#endif
}

#ifndef F2_ENVELOPE_FLAGGED_TU
template <typename F>
using EnvKernel = void (*)(EnvParams, const cpx<F>*);

template <typename F, int... L>
constexpr EnvKernel<F> kernel_table_impl(int log2h, std::integer_sequence<int, L...>) {
    EnvKernel<F> tab[] = {k_envelope<F, L>...};
    return tab[log2h];
}

template <typename F, int MAXL>
EnvKernel<F> kernel_for(int log2h) {
    return kernel_table_impl<F>(log2h, std::make_integer_sequence<int, MAXL + 1>{});
}

constexpr int MAX_LOG2H_F32 = 14;  // rows up to 32768 samples
constexpr int MAX_LOG2H_F64 = 13;  // rows up to 16384 samples

#endif  // !F2_ENVELOPE_FLAGGED_TU

#ifdef F2_ENVELOPE_FLAGGED_TU
// The rows of a length class for launches that exist to serve utterances the spectral kernel's accuracy guard sends back
// (normally none): one workgroup per (utterance, slice of C / ENV_SLICES channels) that leaves at once when its utterance
// is not flagged and otherwise walks its channels. 1 / 16 of k_envelope's workgroups to dispatch when there is nothing to
// do (5 us instead of 75 us per 1000 x 128 rows), and a flagged utterance still spreads over ENV_SLICES workgroups.
constexpr int ENV_SLICES = 8;
template <typename F, int LOG2H>
__global__ __launch_bounds__((threads_for<F, LOG2H>()), (min_waves_for<F, LOG2H>())) void k_envelope_flagged(EnvParams P, const cpx<F>* __restrict__ tw) {
    const int u = blockIdx.x / ENV_SLICES, sl = blockIdx.x - u * ENV_SLICES;
    const int b = P.ulist ? P.ulist[u] : u;
    if (!P.uflag[b]) return;
    const int per = (P.C + ENV_SLICES - 1) / ENV_SLICES;
    for (int c = sl * per; c < min(P.C, (sl + 1) * per); ++c) {
        envelope_row<F, LOG2H>(P, tw, u, c);
        __syncthreads();   // the next row reuses the LDS arrays
    }
}
#endif

}  // namespace

#ifdef F2_ENVELOPE_FLAGGED_TU
int f2_launch_envelope_flagged(f2_ctx* ctx, const f2_env_params& P, int log2h, unsigned nutt) {
    F2_CHECK(ctx, P.uflag && log2h >= F2_SPECTRAL_MIN_LOG2H && log2h <= 14, F2_ERR_INVALID,
             "flagged-row launch outside the spectral kernel's length classes");
    F2_TRY(ensure_twiddles<float>(ctx, log2h, ctx->tw_fl[log2h]));
    const cpx<float>* tw = (const cpx<float>*)ctx->tw_fl[log2h].ptr;
    const dim3 grid(nutt * ENV_SLICES);
    switch (log2h) {
        case 12: hipLaunchKernelGGL((k_envelope_flagged<float, 12>), grid, dim3(threads_for<float, 12>()), 0, ctx->stream, P, tw); break;
        case 13: hipLaunchKernelGGL((k_envelope_flagged<float, 13>), grid, dim3(threads_for<float, 13>()), 0, ctx->stream, P, tw); break;
        default: hipLaunchKernelGGL((k_envelope_flagged<float, 14>), grid, dim3(threads_for<float, 14>()), 0, ctx->stream, P, tw); break;
    }
    F2_HIP(ctx, hipGetLastError());
    return F2_OK;
}
#elif defined(F2_ENVELOPE_P3_TU)
int f2_launch_envelope13_p3(f2_ctx* ctx, const f2_env_params& P, int precision, unsigned rows) {
    constexpr int L13 = 13;
    static_assert(plan_npass(L13) == 3, "this translation unit is compiled with the three-pass plan");
    if (precision == F2_FFT_F32) {
        F2_TRY(ensure_twiddles<float>(ctx, L13, ctx->tw_p3[0]));
        hipLaunchKernelGGL((k_envelope<float, L13>), dim3(rows), dim3(threads_for<float, L13>()), 0, ctx->stream, P,
                           (const cpx<float>*)ctx->tw_p3[0].ptr);
    } else {
        F2_TRY(ensure_twiddles<double>(ctx, L13, ctx->tw_p3[1]));
        hipLaunchKernelGGL((k_envelope<double, L13>), dim3(rows), dim3(threads_for<double, L13>()), 0, ctx->stream, P,
                           (const cpx<double>*)ctx->tw_p3[1].ptr);
    }
    F2_HIP(ctx, hipGetLastError());
    return F2_OK;
}
#else
int f2_launch_envelope(f2_ctx* ctx, const double* d_gfb, const int64_t* d_offsets, const int64_t* h_offsets,
                       int B, int C, int lpf, double cutoff_hz, int precision, double* d_env, const f2_handoff* handoff,
                       const int* d_uflag, const int* h_flag0) {
    const bool f32_in = handoff && handoff->f32;
    F2_CHECK(ctx, !f32_in || precision == F2_FFT_F32, F2_ERR_INVALID, "float32 hand-off needs the float FFT");
    EnvParams P;
    P.f32_in = f32_in ? 1 : 0;
    P.uflag = d_uflag;
    P.stamps = nullptr;
#ifdef F2_STAMPS
    static unsigned long long* d_stamps = nullptr;
    const size_t nstamp = (size_t)B * C * 10;
    if (!d_stamps) F2_HIP(ctx, hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 10 * 128 * 2048));
    if (nstamp <= (size_t)10 * 128 * 2048) P.stamps = d_stamps;
#endif
    P.gfb = d_gfb;
    P.env = d_env;
    P.offsets = d_offsets;
    P.C = C;
    P.lpf = lpf ? 1 : 0;
    // butter(1, cutoff/8000): bilinear transform of 1/(s+1), pre-warped (EnvelopeExtraction.py:47)
    const double k = lpf ? tan(3.14159265358979323846 * cutoff_hz / 16000.0) : 0.0;
    P.b0 = k / (1.0 + k);
    P.a1 = (k - 1.0) / (k + 1.0);

    // group the utterances by padded length (one kernel instantiation per FFT size)
    std::vector<std::vector<int>> groups(32), split_groups(32);
    std::vector<int> pair_group[2];   // [1]: 32769..65536 samples, two sub-rows per workgroup (f2_envelope_pair.hip)
    const bool pair_ok = ctx->opt_env_pair && ((f32_in && handoff->d_x32) || (!f32_in && d_gfb != d_env));
    int n_large = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t n = h_offsets[b + 1] - h_offsets[b];
        if (n <= 0) continue;
        const int log2m = n <= 2 ? 1 : f2_log2_ceil(n);
        const int log2h = log2m - 1;
        const int maxl = precision == F2_FFT_F32 ? MAX_LOG2H_F32 : MAX_LOG2H_F64;
        if (pair_ok && f2_envelope_pair_supports(log2h, precision) && (!f32_in || (handoff->h_x32_off && handoff->h_x32_off[b] >= 0))) {
            pair_group[log2h - 14].push_back(b);
            ++n_large;
            continue;
        }
        if (log2h > maxl && f2_envelope_split_supports(log2h, precision)) {
            // longer than the LDS-resident transform: four-step transform, all utterances of a size together
            split_groups[log2h].push_back(b);
            ++n_large;
            continue;
        }
        if (log2h > maxl) {
            // beyond that (or float64 transforms): global-memory radix-16 passes, one utterance at a time
            F2_CHECK(ctx, !f32_in, F2_ERR_INVALID, "float32 hand-off is not available for long rows");
            F2_TRY(f2_prof_begin(ctx, F2_K_ENVELOPE));
            F2_TRY(f2_launch_envelope_large(ctx, d_gfb + (size_t)C * (size_t)h_offsets[b],
                                            d_env + (size_t)C * (size_t)h_offsets[b], n, C, P.lpf, P.b0, P.a1, precision));
            F2_TRY(f2_prof_end(ctx, F2_K_ENVELOPE));
            ++n_large;
            continue;
        }
        groups[log2h].push_back(b);
    }
    for (int g = 0; g < 2; ++g)
        if (!pair_group[g].empty())
            F2_TRY(f2_launch_envelope_pair(ctx, d_gfb, d_env, d_offsets, h_offsets, pair_group[g].data(),
                                           (int)pair_group[g].size(), 14 + g, C, P.lpf, P.b0, P.a1,
                                           f32_in ? handoff->d_x32 : nullptr, f32_in ? handoff->d_x32_off : nullptr,
                                           f32_in ? handoff->h_x32_off : nullptr, d_uflag));
    for (int log2h = 0; log2h < 32; ++log2h)
        if (!split_groups[log2h].empty())
            F2_TRY(f2_launch_envelope_split(ctx, d_gfb, d_env, d_offsets, split_groups[log2h].data(),
                                            (int)split_groups[log2h].size(), log2h, C, P.lpf, P.b0, P.a1,
                                            f32_in ? handoff->d_x32 : nullptr, f32_in ? handoff->d_x32_off : nullptr, d_uflag));
    // Utterances of the spectral kernel's length classes that the HOST kept off that route (flag 1 from the start: too little
    // padding for its accuracy guard) are known here: they go first in their group and get one workgroup per row like any
    // envelope launch; the launch that walks the flags with 1 / 16 of the workgroups serves the rest, which only the guard can
    // send back. (All of them behind that launch took 0.9 + 0.5 ms per 2500-utterance launch of the ragged corpus: sixteen rows
    // one after the other on the few workgroups that own an active utterance.)
    std::vector<int> nactive(32, 0);
    bool reordered = false;
    if (d_uflag && h_flag0 && precision == F2_FFT_F32)
        for (int l = F2_SPECTRAL_MIN_LOG2H; l <= 14; ++l) {
            auto& g = groups[l];
            const auto mid = std::stable_partition(g.begin(), g.end(), [&](int b) { return h_flag0[b] != 0; });
            nactive[l] = (int)(mid - g.begin());
            reordered = reordered || (nactive[l] > 0 && nactive[l] < (int)g.size());
        }
    size_t list_elems = 0;
    int ngroups = 0;
    for (auto& g : groups)
        if (!g.empty()) {
            ++ngroups;
            list_elems += g.size();
        }
    const bool identity = ngroups == 1 && (int)list_elems == B && n_large == 0 && !reordered;
    int* d_lists = nullptr;
    if (!identity && ngroups > 0) {
        std::vector<int> flat;
        flat.reserve(list_elems);
        for (auto& g : groups) flat.insert(flat.end(), g.begin(), g.end());
        F2_TRY(f2_reserve(ctx, ctx->work2, sizeof(int) * flat.size()));
        F2_TRY(f2_upload_async(ctx, ctx->work2.ptr, flat.data(), sizeof(int) * flat.size()));
        d_lists = (int*)ctx->work2.ptr;
    }
    size_t pos = 0;
    for (int log2h = 0; log2h < 32; ++log2h) {
        const auto& g = groups[log2h];
        if (g.empty()) continue;
        const int* list = identity ? nullptr : d_lists + pos;
        pos += g.size();
        const int nthreads = (log2h == 14 && precision == F2_FFT_F32) ? 1024
                             : (log2h == 13 && precision == F2_FFT_F32) ? F2_THREADS13 : log2h >= 13 ? 512 : 256;   // threads_for<F, LOG2H>()
        // one workgroup per row for the utterances ulist[0 .. count)
        auto launch_rows = [&](const int* ulist, size_t count) -> int {
            P.ulist = ulist;
            const dim3 grid((unsigned)(count * (size_t)C)), block(nthreads);
            if (log2h == 13 && (!P.lpf || precision == F2_FFT_F64) && !ctx->opt_env_plan4) {
                // the 1 s row without the float low-pass: three-pass plan (f2_envelope_p3.hip)
                F2_TRY(f2_launch_envelope13_p3(ctx, P, precision, (unsigned)(count * (size_t)C)));
            } else if (precision == F2_FFT_F32) {
                F2_TRY(ensure_twiddles<float>(ctx, log2h, ctx->tw[0][log2h]));
                const EnvKernel<float> kern = kernel_for<float, MAX_LOG2H_F32>(log2h);
                hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, P, (const cpx<float>*)ctx->tw[0][log2h].ptr);
            } else {
                F2_TRY(ensure_twiddles<double>(ctx, log2h, ctx->tw[1][log2h]));
                const EnvKernel<double> kern = kernel_for<double, MAX_LOG2H_F64>(log2h);
                hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, P, (const cpx<double>*)ctx->tw[1][log2h].ptr);
            }
            return F2_OK;
        };
        F2_TRY(f2_prof_begin(ctx, F2_K_ENVELOPE));
        if (precision == F2_FFT_F32 && P.uflag && log2h >= F2_SPECTRAL_MIN_LOG2H && log2h <= 14) {
            // a length class the spectral kernel serves: the utterances the host routed here first, one workgroup per row;
            // the others only come here when the guard flags them - walked by far fewer workgroups (f2_envelope_flagged.hip)
            const size_t na = (size_t)nactive[log2h];
            if (na > 0) F2_TRY(launch_rows(list, na));
            if (na < g.size()) {
                P.ulist = list ? list + na : nullptr;
                F2_TRY(f2_launch_envelope_flagged(ctx, P, log2h, (unsigned)(g.size() - na)));
            }
        } else {
            F2_TRY(launch_rows(list, g.size()));
        }
        F2_HIP(ctx, hipGetLastError());
        F2_TRY(f2_prof_end(ctx, F2_K_ENVELOPE));
    }
#ifdef F2_STAMPS
    if (P.stamps) {
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<unsigned long long> h(nstamp);
        F2_HIP(ctx, hipMemcpy(h.data(), d_stamps, sizeof(unsigned long long) * nstamp, hipMemcpyDeviceToHost));
        double acc[10] = {0};
        const size_t rows = (size_t)B * C;
        for (size_t r = 0; r < rows; ++r)
            for (int k = 1; k < 10; ++k) acc[k] += (double)(h[r * 10 + k] - h[r * 10 + k - 1]);
        static const char* names[10] = {"", "load", "fwd fft", "hilbert pairs", "inv fft", "magnitude", "transposed write",
                                        "lpf chunk+scan", "lpf carry+apply", "copy out"};
        fprintf(stderr, "[stamps] mean cycles per workgroup (s_memrealtime, 10 ns ticks):");
        for (int k = 1; k < 10; ++k) fprintf(stderr, " %s=%.0f", names[k], acc[k] / rows);
        fprintf(stderr, "\n");
        // residency: sum of workgroup lifetimes / kernel span = workgroups alive at once (chip-wide)
        unsigned long long t0 = ~0ull, t1 = 0;
        double life = 0;
        for (size_t r = 0; r < rows; ++r) {
            t0 = std::min(t0, h[r * 10]);
            t1 = std::max(t1, h[r * 10 + 9]);
            life += (double)(h[r * 10 + 9] - h[r * 10]);
        }
        fprintf(stderr, "[stamps] mean workgroup lifetime %.1f ticks, kernel span %.0f ticks, workgroups alive at once %.1f\n",
                life / rows, (double)(t1 - t0), life / (double)(t1 - t0));
    }
#endif
    return F2_OK;
}
#endif  // F2_ENVELOPE_P3_TU
