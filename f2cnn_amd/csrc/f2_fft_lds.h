// LDS-resident Stockham FFT machinery shared by the envelope kernels: radix plans, per-pass twiddle tables, one pass
// (fft_pass) and whole transforms (fft_all, fft_regs_to_regs). See f2_envelope.hip for how K2 uses it.
#pragma once
#include <cmath>
#include <vector>

#include "f2_envelope_core.h"

namespace f2fft {

// knock-out (timing only, results wrong): -DF2_KO_BARRIER turns the workgroup barriers of the transforms into nothing
#ifdef F2_KO_BARRIER
#define F2_FFT_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define F2_FFT_BARRIER() __syncthreads()
#endif

#ifndef F2_PLAN13_PASSES
#define F2_PLAN13_PASSES 4
#endif
#ifndef F2_THREADS13       // workgroup size of the H = 8192 float transform (diagnostic variants: 256)
#define F2_THREADS13 512
#endif
#ifndef F2_MINWAVES13
#define F2_MINWAVES13 4
#endif
// threads per workgroup: 256, or 512 where 256 threads would need more than 256 registers each
template <typename F, int LOG2H>
constexpr int threads_for() {
    return (LOG2H == 14 && sizeof(F) == 4) ? 1024 : (LOG2H == 13 && sizeof(F) == 4) ? F2_THREADS13 : LOG2H >= 13 ? 512 : 256;
}
// waves per SIMD the register allocator must leave room for (2 workgroups per CU wherever LDS allows)
template <typename F, int LOG2H>
constexpr int min_waves_for() {
    return (sizeof(F) == 4 && LOG2H == 13) ? F2_MINWAVES13 : (sizeof(F) == 4 && LOG2H == 14) ? 4 : 2;
}

// Whether pass 0 derives its 15 twiddles per butterfly from two loaded ones (radix-16 first pass, float transforms)
template <typename F, int LOG2H>
constexpr bool derive_tw0() { return sizeof(F) == 4 && LOG2H >= 11 && LOG2H <= 14; }

// Whether the Hilbert pair step is folded into the inverse transform's first pass (needs ~3x the pass's points in
// registers for a moment: only where the register budget allows) or runs as its own sweep over LDS.
#ifndef F2_FUSE_HILBERT_MAX
#define F2_FUSE_HILBERT_MAX 12
#endif
template <typename F, int LOG2H>
constexpr bool fuse_hilbert() { return LOG2H >= 1 && LOG2H <= F2_FUSE_HILBERT_MAX && sizeof(F) == 4; }

// ---- radix plan: symmetric (first radix == last radix), radices 2..32 ----
// H = 8192 (the 1 s / 16 kHz row) runs as 16-8-4-16 on 512 threads: 16 points per thread in every pass keeps
// the kernel under 128 registers, i.e. 16 waves per CU to hide LDS / barrier / HBM latency. H = 16384 does the same
// with 16-16-4-16 on 1024 threads (one workgroup per CU by LDS, still 16 waves).
constexpr int plan_npass(int h) {
    return h == 0 ? 0 : h <= 5 ? 1 : h <= 10 ? ((h & 1) ? 3 : 2) : (h == 13 && F2_PLAN13_PASSES == 4) ? 4 : h == 14 ? 4 : 3;
}
constexpr int plan_bits(int h, int pass) {
    if (h <= 5) return h;
    if (h <= 10) return (h & 1) ? (pass == 1 ? 1 : (h - 1) / 2) : h / 2;
    if (h == 13 && F2_PLAN13_PASSES == 4) return pass == 1 ? 3 : pass == 2 ? 2 : 4;
    if (h == 14) return pass == 2 ? 2 : 4;   // 16-16-4-16 on 1024 threads: 16 points per thread, as for h == 13
    const int a = h <= 13 ? 4 : 5;
    return pass == 1 ? h - 2 * a : a;
}
// most complex points a thread holds in any pass
constexpr int plan_points_per_thread(int h, int nt) {
    int m = 1;
    for (int p = 0; p < plan_npass(h); ++p) {
        const int R = 1 << plan_bits(h, p);
        const int nb = (1 << h) / R;
        const int pts = ((nb + nt - 1) / nt) * R;
        m = pts > m ? pts : m;
    }
    return m;
}
constexpr int plan_shift(int h, int pass) {  // log2 of the stride entering `pass`
    int s = 0;
    for (int i = 0; i < pass; ++i) s += plan_bits(h, i);
    return s;
}
// per-pass twiddle table: entry (k-1)*(NB/S) + p holds exp(-2 pi i (p*S) k / H), p < NB/S, 1 <= k < R
constexpr int plan_tw_count(int h, int pass) {
    const int R = 1 << plan_bits(h, pass), S = 1 << plan_shift(h, pass), H = 1 << h;
    return S * R == H ? 0 : (R - 1) * (H / R / S);
}
constexpr int plan_tw_offset(int h, int pass) {
    int o = 0;
    for (int i = 0; i < pass; ++i) o += plan_tw_count(h, i);
    return o;
}
constexpr int plan_tw_total(int h) { return plan_tw_offset(h, plan_npass(h)); }
// The tables of passes >= 1 are small (a pass with stride S has only NB/S distinct twiddle columns): each workgroup
// copies them to LDS once, so only pass 0 (one distinct column per butterfly) reads its twiddles from global memory.
constexpr int plan_tw_lds_count(int h) { return plan_npass(h) >= 2 ? plan_tw_total(h) - plan_tw_offset(h, 1) : 0; }   // then H/2+1 entries of V

// LDS index padding of the complex array: one extra slot per 16
__device__ __forceinline__ int cpad(int i) { return i + (i >> 4); }
constexpr int cpad_size(int h) { return h + (h >> 4) + 1; }

// One Stockham pass. SRC_REGS: inputs are already in v (first pass, loaded from global memory);
// DST_REGS: outputs stay in v (last pass of the inverse transform). v is indexed [i*R + j] with
// butterfly bf = tid + i*NT and point bf + j*NB.
// HILBERT (first pass of the inverse transform): the pass reads Z[m] and its mirror Z[H-m] and forms, in
// registers, the conjugated and 1/H-scaled packed spectrum of the Hilbert transform
//   W[m] = i sin(t_m) Z[m] + cos(t_m) conj(Z[H-m]),  t_m = 2 pi m / M,  W[0] = 0
// so the spectrum never makes a separate trip through LDS.
template <typename F, int LOG2H, int PASS, bool SRC_REGS, bool DST_REGS, int PTV, int NT, bool HILBERT = false,
          bool T0REGS = false>
__device__ __forceinline__ void fft_pass(cpx<F>* lds, const cpx<F>* __restrict__ tw, const cpx<F>* twl, int tid,
                                         cpx<F> (&v)[PTV]) {
    constexpr int H = 1 << LOG2H;
    constexpr int R = 1 << plan_bits(LOG2H, PASS);
    constexpr int LOG2S = plan_shift(LOG2H, PASS);
    constexpr int S = 1 << LOG2S;
    constexpr int NB = H / R;
    constexpr int ITER = (NB + NT - 1) / NT;
    constexpr bool LAST = (S * R == H);
    static_assert(ITER * R <= PTV, "register array too small");
    constexpr bool FULL = NB % NT == 0;   // every thread owns ITER whole butterflies: no guards
    const cpx<F>* __restrict__ twp = tw + plan_tw_offset(LOG2H, PASS);
#ifdef F2_KO_X12   // knock-out (timing only): no LDS exchange between passes 1 and 2
    constexpr bool KO_READ = PASS == 2, KO_WRITE = PASS == 1;
#else
    constexpr bool KO_READ = false, KO_WRITE = false;
#endif
    if constexpr (!SRC_REGS && !KO_READ) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int bf = tid + i * NT;
#ifdef F2_KO_LDS
            if constexpr (false) {
#else
            if (FULL || bf < NB) {
#endif
                // cpad(bf + j*NB) = cpad(bf) + j*(NB + NB/16) when 16 | NB: one base + immediate offsets
                if constexpr (NB % 16 == 0) {
                    const cpx<F>* src = lds + cpad(bf);
#pragma unroll
                    for (int j = 0; j < R; ++j) v[i * R + j] = src[j * (NB + NB / 16)];
                } else {
#pragma unroll
                    for (int j = 0; j < R; ++j) v[i * R + j] = lds[cpad(bf + j * NB)];
                }
                if constexpr (HILBERT) {
                    const cpx<F>* __restrict__ V = tw + plan_tw_total(LOG2H);   // (cos t_k, -sin t_k), k <= H/2
                    const F sc = F(1.0 / H);
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const int m = bf + j * NB;
                        // mirror point; cpad(H - m) = cpad(H - bf) - j*(NB + NB/16) when 16 | NB
                        const cpx<F> zp = (NB % 16 == 0) ? (lds + cpad(H - bf))[-j * (NB + NB / 16)] : lds[cpad(H - m)];
                        const bool upper = 2 * m > H;                  // t_m = pi - t_(H-m)
                        const cpx<F> vk = V[upper ? H - m : m];
                        const F cs = (upper ? -vk.re : vk.re) * sc, sn = -vk.im * sc;
                        const cpx<F> z = v[i * R + j];
                        const F wre = cs * zp.re - sn * z.im, wim = sn * z.re - cs * zp.im;
                        v[i * R + j] = {m == 0 ? F(0) : wre, m == 0 ? F(0) : -wim};
                    }
                }
            }
        }
        F2_FFT_BARRIER();
    }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int bf = tid + i * NT;
        if (FULL || bf < NB) {
#ifndef F2_KO_DFT   // knock-outs: timing experiments only (tools/build_variant.sh), results are wrong
            dft<R>(&v[i * R]);   // X[k] now sits in v[i*R + brev<R>(k)]
#endif
#ifdef F2_KO_TW   // bit mask of passes that skip their twiddles
            if constexpr (!LAST && !((F2_KO_TW >> PASS) & 1)) {
#else
            if constexpr (!LAST) {
#endif
                const cpx<F>* twq = (PASS >= 1 ? twl + (plan_tw_offset(LOG2H, PASS) - plan_tw_offset(LOG2H, 1)) : twp) +
                                    (bf >> LOG2S);
                if constexpr (PASS == 0 && T0REGS && R == 16) {
                    // pass 0 has one distinct twiddle column per butterfly: load w and w^4 (two coalesced loads from
                    // an 8 KB-per-workgroup slice that stays in L1) and form the other 13 powers by products,
                    // at most three roundings deep, instead of 15 loads from a 60 KB table
                    cpx<F> w[R];
                    w[1] = twq[0];
                    w[4] = twq[3 * (NB / S)];
                    w[2] = cmul(w[1], w[1]);
                    w[3] = cmul(w[2], w[1]);
                    w[8] = cmul(w[4], w[4]);
                    w[5] = cmul(w[4], w[1]);
                    w[6] = cmul(w[4], w[2]);
                    w[7] = cmul(w[4], w[3]);
                    w[12] = cmul(w[8], w[4]);
                    w[9] = cmul(w[8], w[1]);
                    w[10] = cmul(w[8], w[2]);
                    w[11] = cmul(w[8], w[3]);
                    w[13] = cmul(w[12], w[1]);
                    w[14] = cmul(w[12], w[2]);
                    w[15] = cmul(w[12], w[3]);
#pragma unroll
                    for (int k = 1; k < R; ++k) v[i * R + brev<R>(k)] = cmul(v[i * R + brev<R>(k)], w[k]);
                } else {
#pragma unroll
                    for (int k = 1; k < R; ++k)
                        v[i * R + brev<R>(k)] = cmul(v[i * R + brev<R>(k)], twq[(k - 1) * (NB / S)]);
                }
            }
#ifdef F2_KO_LDS
            if constexpr (false) {
#else
            if constexpr (!DST_REGS && !KO_WRITE) {
#endif
                const int q = bf & (S - 1);
                const int base = q + (bf - q) * R;
                if constexpr (S % 16 == 0) {
                    cpx<F>* dst = lds + cpad(base);
#pragma unroll
                    for (int k = 0; k < R; ++k) dst[k * (S + S / 16)] = v[i * R + brev<R>(k)];
                } else {
#pragma unroll
                    for (int k = 0; k < R; ++k) lds[cpad(base + S * k)] = v[i * R + brev<R>(k)];
                }
            }
        }
    }
    if constexpr (!DST_REGS && !KO_WRITE) F2_FFT_BARRIER();
}

template <typename F, int LOG2H, bool INVERSE, int PTV, int NT, bool T0REGS = false, int PASS = 0>
__device__ __forceinline__ void fft_all(cpx<F>* lds, const cpx<F>* __restrict__ tw, const cpx<F>* twl, int tid,
                                        cpx<F> (&v)[PTV]) {
    constexpr int NP = plan_npass(LOG2H);
    if constexpr (PASS < NP) {
        constexpr bool SRC = !INVERSE && PASS == 0;
        constexpr bool DST = INVERSE && PASS == NP - 1;
        fft_pass<F, LOG2H, PASS, SRC, DST, PTV, NT, INVERSE && PASS == 0 && fuse_hilbert<F, LOG2H>(), T0REGS>(lds, tw, twl, tid, v);
        fft_all<F, LOG2H, INVERSE, PTV, NT, T0REGS, PASS + 1>(lds, tw, twl, tid, v);
    }
}


// A whole transform whose inputs are in v (first-pass layout) and whose outputs return to v (same layout as the
// inverse transform of fft_all leaves them): used where a spectrum is formed in registers between two transforms.
template <typename F, int LOG2H, int PTV, int NT, bool T0REGS = false, int PASS = 0>
__device__ __forceinline__ void fft_regs_to_regs(cpx<F>* lds, const cpx<F>* __restrict__ tw, const cpx<F>* twl, int tid,
                                                 cpx<F> (&v)[PTV]) {
    constexpr int NP = plan_npass(LOG2H);
    if constexpr (PASS < NP) {
        fft_pass<F, LOG2H, PASS, PASS == 0, PASS == NP - 1, PTV, NT, false, T0REGS>(lds, tw, twl, tid, v);
        fft_regs_to_regs<F, LOG2H, PTV, NT, T0REGS, PASS + 1>(lds, tw, twl, tid, v);
    }
}

// host: per-pass twiddle tables followed by V[k] = exp(-2 pi i k / M), k <= H/2 (long double trigonometry)
template <typename F>
int ensure_twiddles(f2_ctx* ctx, int log2h, f2_scratch& slot) {
    if (slot.ptr) return F2_OK;
    const int H = 1 << log2h;
    const long double tau = 2.0L * 3.14159265358979323846264338327950288L;
    std::vector<cpx<F>> host;
    host.reserve((size_t)plan_tw_total(log2h) + H / 2 + 1);
    for (int pass = 0; pass < plan_npass(log2h); ++pass) {
        const int R = 1 << plan_bits(log2h, pass), S = 1 << plan_shift(log2h, pass);
        if (S * R == H) continue;
        const int np = H / R / S;
        for (int k = 1; k < R; ++k)
            for (int p = 0; p < np; ++p) {
                const long double ang = tau * (long double)((int64_t)p * S * k % H) / (long double)H;
                host.push_back({(F)cosl(ang), (F)(-sinl(ang))});
            }
    }
    if ((int)host.size() != plan_tw_total(log2h)) return f2_fail(ctx, F2_ERR_INVALID, "twiddle plan mismatch");
    for (int k = 0; k <= H / 2; ++k) {
        const long double ang = tau * (long double)k / (long double)(2 * H);
        host.push_back({(F)cosl(ang), (F)(-sinl(ang))});
    }
    F2_TRY(f2_reserve(ctx, slot, sizeof(cpx<F>) * host.size()));
    F2_TRY(f2_upload_async(ctx, slot.ptr, host.data(), sizeof(cpx<F>) * host.size()));   // (a local: staged, not waited for)
    return F2_OK;
}


}  // namespace f2fft
