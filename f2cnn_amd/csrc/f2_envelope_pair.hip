// K2 for rows of 32769..65536 samples (most TIMIT sentences: 2-4 s at 16 kHz), float transforms: the whole row stays on
// the chip. Same mathematics as f2_envelope.hip (reference: scripts/processing/EnvelopeExtraction.py:20-67).
//
// M = 65536 needs two complex transforms of H = 32768 points = 256 KB, more than the 160 KB of LDS. One radix-2 stage
// at each end splits them into two INDEPENDENT sub-rows of HS = 16384 points that the LDS-resident machinery of
// f2_fft_lds.h transforms one after the other inside the same 1024-thread workgroup:
//
//   load      z[m] = x[2m] + i x[2m+1];  a[m] = z[m] + z[m+HS],  b[m] = (z[m] - z[m+HS]) e^{-2 pi i m / H}   (DIF stage)
//             => Z[2j] = FFT_HS(a)[j],  Z[2j+1] = FFT_HS(b)[j]
//   pair step W[k] = i sin(t_k) Z[k] + cos(t_k) conj(Z[H-k]),  t_k = pi k / H,  W[0] = 0  (as f2_envelope.hip) never
//             mixes parities: H - 2j = 2 (HS - j) and H - (2j+1) = 2 (HS-1-j) + 1. The even sub-row is exactly the pair
//             step of an HS-point row (t = pi j / HS); the odd one pairs j with HS-1-j at t = pi (2j+1) / H
//   inverse   conj(w[m]) = FFT_H(conj(W)/H)[m] = (E[m] + e^{-2 pi i m / H} O[m]) / 2,  conj(w[m+HS]) = (E[m] - ... O[m]) / 2
//             with E, O = FFT_HS of the two conjugated, 1/HS-scaled sub-spectra (DIT stage); only squares of w are used
//
// While one sub-row is in LDS the other one (b, then E) is parked in the row's own float64 OUTPUT slot - 8 n >= 2 * 128 KB
// bytes that nothing reads before this workgroup overwrites them with the envelopes, and that stay in L2 / Infinity Cache
// for the ~100 us in between. This requires the input not to live there: the filterbank's float32 hand-off of such rows
// is a compact scratch row anyway (f2_plan_handoff), and float64 input qualifies when gfb != env; in-place float64 calls
// keep the four-step path (f2_envelope_split.hip). HBM bytes per sample-channel: 4 (or 8) read twice + 8 written,
// against 44-48 for the four-step path.
#include <algorithm>

#include "f2_fft_lds.h"

using namespace f2fft;

namespace {

// Sub-row geometry: LOG2S = 14 (rows of 32769..65536 samples, 1024 threads, one workgroup per CU); the kernel is generic
// in it (13: 512 threads, two workgroups per CU - measured slower than the single-row plan, see below).
template <int LOG2S>
struct Geo {
    static constexpr int HS = 1 << LOG2S;
    static constexpr int NT = threads_for<float, LOG2S>();
    static constexpr int PT = plan_points_per_thread(LOG2S, NT);
    static constexpr int R0 = 1 << plan_bits(LOG2S, 0);
    static constexpr int NB0 = HS / R0;
    static_assert(PT == R0 && NB0 == NT, "one radix-16 butterfly per thread in the first and last pass");
};
#ifndef F2_PAIR_LOADCHUNK
#define F2_PAIR_LOADCHUNK 8
#endif
constexpr int LOADCHUNK = F2_PAIR_LOADCHUNK;   // points whose loads are in flight together in the two radix-2 stages

struct PairParams {
    const double* gfb;
    double* env;
    const int64_t* offsets;
    const int* ulist;          // utterances of this launch (device)
    const int* uflag;          // per utterance of the batch, or NULL: utterances whose flag is 0 are left alone
    int C;
    int lpf;
    double b0, a1;
    const float* x32;          // float32 hand-off of the filterbank: compact (C, n) rows at x32 + x32_off[b], or NULL
    const int64_t* x32_off;
    const cpx<float>* tx;      // [HS]   exp(-2 pi i m / H) = (cos, -sin)(pi m / HS)
    const cpx<float>* vo;      // [HS/2] (cos, -sin)(pi (2j+1) / H): pair-step angles of the odd sub-row
    unsigned long long* stamps;   // diagnostic build only (-DF2_STAMPS)
};

// Diagnostic build only (-DF2_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at the phase boundaries.
#ifdef F2_STAMPS
#define F2_PSTAMP(k)                                         \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        pst[k] = __builtin_amdgcn_s_memrealtime();             \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#else
#define F2_PSTAMP(k) \
    do {             \
    } while (0)
#endif

// samples i0, i0 + 1 (i0 even) of a row of n samples as floats, 0 beyond the end
template <typename T>
__device__ __forceinline__ cpx<float> load_pair(const T* __restrict__ x, int n, int i0) {
    cpx<float> r;
    row_pair<true>(x, n, i0, r.re, r.im);
    return r;
}

// forward transform of the sub-row in v, Hilbert pair step in LDS, second transform back into v
template <int LOG2S, bool ODD>
__device__ __forceinline__ void sub_row(cpx<float>* lds, const cpx<float>* __restrict__ tw, const cpx<float>* twl,
                                        const cpx<float>* __restrict__ vo, int tid, cpx<float> (&v)[Geo<LOG2S>::PT]) {
    constexpr int HS = Geo<LOG2S>::HS, NT = Geo<LOG2S>::NT, PT = Geo<LOG2S>::PT;
    constexpr bool T0R = derive_tw0<float, LOG2S>();
    asm volatile("" : "+v"(tid));   // (nothing derived from the thread index outside this sub-row is kept alive through it)
    fft_all<float, LOG2S, false, PT, NT, T0R>(lds, tw, twl, tid, v);       // spectrum at lds[cpad(k)]
    const float sc = 1.0f / (float)HS;
    if constexpr (!ODD) {
        const cpx<float>* __restrict__ V = tw + plan_tw_total(LOG2S);      // (cos, -sin)(pi k / HS), k <= HS/2
        constexpr int NPAIR = HS / 2 - 1, ITERP = (NPAIR + NT - 1) / NT;
        cpx<float> zk[ITERP], zh[ITERP], vk[ITERP];
#pragma unroll
        for (int i = 0; i < ITERP; ++i) {
            const int k = min(1 + tid + i * NT, HS / 2 - 1);
            zk[i] = lds[cpad(k)];
            zh[i] = lds[cpad(HS - k)];
            vk[i] = V[k];
        }
#pragma unroll
        for (int i = 0; i < ITERP; ++i) {
            const int k = 1 + tid + i * NT;
            const float cs = vk[i].re * sc, sn = -vk[i].im * sc;
            if (k < HS / 2) {
                lds[cpad(k)] = {-sn * zk[i].im + cs * zh[i].re, -(sn * zk[i].re - cs * zh[i].im)};
                lds[cpad(HS - k)] = {-sn * zh[i].im - cs * zk[i].re, -(sn * zh[i].re + cs * zk[i].im)};
            }
        }
        if (tid == 0) {
            const cpx<float> zq = lds[cpad(HS / 2)];                       // t = pi/2: W = i Z
            lds[cpad(HS / 2)] = {-zq.im * sc, -zq.re * sc};
            lds[cpad(0)] = {0.f, 0.f};
        }
    } else {
        constexpr int ITERP = (HS / 2) / NT;                               // pairs (j, HS-1-j), j < HS/2
        cpx<float> zk[ITERP], zh[ITERP], vk[ITERP];
#pragma unroll
        for (int i = 0; i < ITERP; ++i) {
            const int j = tid + i * NT;
            zk[i] = lds[cpad(j)];
            zh[i] = lds[cpad(HS - 1 - j)];
            vk[i] = vo[j];
        }
#pragma unroll
        for (int i = 0; i < ITERP; ++i) {
            const int j = tid + i * NT;
            const float cs = vk[i].re * sc, sn = -vk[i].im * sc;
            lds[cpad(j)] = {-sn * zk[i].im + cs * zh[i].re, -(sn * zk[i].re - cs * zh[i].im)};
            lds[cpad(HS - 1 - j)] = {-sn * zh[i].im - cs * zk[i].re, -(sn * zh[i].re + cs * zk[i].im)};
        }
    }
    __syncthreads();
    fft_all<float, LOG2S, true, PT, NT, T0R>(lds, tw, twl, tid, v);        // point tid + j*NB0 in v[brev<R0>(j)]
}

// T: float (the filterbank's compact float32 hand-off rows) or double (a float64 filterbank matrix)
template <int LOG2S, typename T>
__global__ __launch_bounds__((Geo<LOG2S>::NT), 4) void k_envelope_pair(PairParams P, const cpx<float>* __restrict__ tw) {
    constexpr int HS = Geo<LOG2S>::HS, NT = Geo<LOG2S>::NT, PT = Geo<LOG2S>::PT, R0 = Geo<LOG2S>::R0, NB0 = Geo<LOG2S>::NB0;
    constexpr int CS = cpad_size(HS);
    constexpr int TWL = plan_tw_lds_count(LOG2S);
    constexpr size_t LP = lowpass_lds_bytes<float, NT, R0>();
    constexpr size_t LDS_BYTES = sizeof(cpx<float>) * CS > LP ? sizeof(cpx<float>) * CS : LP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    __shared__ __attribute__((aligned(16))) cpx<float> twl[TWL > 0 ? TWL : 1];
    __shared__ float e_mid;

    const int tid = threadIdx.x;
    const int u = blockIdx.x / P.C, c = blockIdx.x - u * P.C;
    const int b = P.ulist[u];
    if (P.uflag && !P.uflag[b]) return;   // (whole workgroup) served by the spectral kernel
    const int64_t off = P.offsets[b];
    const int n = (int)(P.offsets[b + 1] - off);
    const size_t row = (size_t)P.C * (size_t)off + (size_t)c * (size_t)n;
    for (int i = tid; i < TWL; i += NT) twl[i] = tw[plan_tw_offset(LOG2S, 1) + i];
    double* __restrict__ y = P.env + row;
    const T* __restrict__ x;
    if constexpr (sizeof(T) == 4)
        x = reinterpret_cast<const T*>(P.x32 + P.x32_off[b] + (size_t)c * (size_t)n);
    else
        x = reinterpret_cast<const T*>(P.gfb + row);
    float* e_mid_p = &e_mid;
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem);
    cpx<float>* park_b = reinterpret_cast<cpx<float>*>(y);      // b: [0, 128 KB) of the row's output slot
    cpx<float>* park_e = park_b + HS;                           // E: [128 KB, 256 KB)

    // load + first radix-2 stage; point j of this thread is m = tid + j * NB0. A few points at a time: with all 16
    // in flight the loads alone would hold 96 registers.
    cpx<float> v[PT];
#ifdef F2_STAMPS
    unsigned long long pst[8] = {0};
#endif
    F2_PSTAMP(0);
#pragma unroll
    for (int j0 = 0; j0 < R0; j0 += LOADCHUNK) {
#pragma unroll
        for (int j = j0; j < j0 + LOADCHUNK; ++j) {
            int m = tid + j * NB0;
            asm volatile("" : "+v"(m));       // (addresses are recomputed per stage, not held in registers across the transforms)
            const cpx<float> z0 = load_pair<T>(x, n, 2 * m), z1 = load_pair<T>(x, n, 2 * (m + HS));
            v[j] = z0 + z1;
            park_b[m] = cmul(z0 - z1, P.tx[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    F2_PSTAMP(1);
    sub_row<LOG2S, false>(lds, tw, twl, P.vo, tid, v);
    F2_PSTAMP(2);
#pragma unroll
    for (int j = 0; j < R0; ++j) {
        int m = tid + j * NB0;
        asm volatile("" : "+v"(m));
        park_e[m] = v[brev<R0>(j)];
    }
#pragma unroll
    for (int j = 0; j < R0; ++j) {
        int m = tid + j * NB0;
        asm volatile("" : "+v"(m));
        v[j] = park_b[m];       // this thread's own stores: no synchronisation
    }
    F2_PSTAMP(3);
    sub_row<LOG2S, true>(lds, tw, twl, P.vo, tid, v);
    F2_PSTAMP(4);

    // last radix-2 stage + magnitude: lower half of the row from E + T O, upper half from E - T O. The lower half's
    // envelopes stay in registers until every thread has read its parked points (they overwrite the parking area);
    // the upper half's positions lie beyond it and are written at once - final float64 values without low-pass, else
    // the float pair at the start of its own 16-byte output position, where it waits for the second low-pass segment.
    // (Packing those pairs into whole lines at the start of the upper half's region instead: 8 % faster for n = 65536,
    // 2-3 % slower for n <= 40000, 1.7 % slower on the U[1 s, 4 s] corpus: not kept.)
    float er[R0], ei[R0];
#pragma unroll
    for (int j0 = 0; j0 < R0; j0 += 4) {
#pragma unroll
        for (int j = j0; j < j0 + 4; ++j) {
            int m = tid + j * NB0;
            asm volatile("" : "+v"(m));       // (keeps the compiler from holding the load stage's 32 addresses until here)
            const cpx<float> e = park_e[m];
            const cpx<float> o = cmul(v[brev<R0>(j)], P.tx[m]);
            const cpx<float> x0 = load_pair<T>(x, n, 2 * m), x1 = load_pair<T>(x, n, 2 * (m + HS));
            const float lr = 0.5f * (e.re + o.re), li = 0.5f * (e.im + o.im);
            const float ur = 0.5f * (e.re - o.re), ui = 0.5f * (e.im - o.im);
            er[j] = fsqrt(x0.re * x0.re + lr * lr);
            ei[j] = fsqrt(x0.im * x0.im + li * li);
            const float hr = fsqrt(x1.re * x1.re + ur * ur), hi = fsqrt(x1.im * x1.im + ui * ui);
            const int i1 = 2 * (m + HS);
            if (P.lpf) {
                if (i1 < n) *reinterpret_cast<cpx<float>*>(y + i1) = {hr, hi};
            } else {
                store_row_pair(y, n, i1, (double)hr, (double)hi);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == NT - 1) *e_mid_p = ei[R0 - 1];   // envelope sample 2 HS - 1: e[n-1] entering the upper half's low-pass
    F2_PSTAMP(5);
    __syncthreads();   // every read of the parked sub-rows (and the last LDS reads) precedes the stores over them
    if (!P.lpf) {
#pragma unroll
        for (int j = 0; j < R0; ++j) {
            store_pair(y + 2 * (tid + j * NB0), (double)er[j], (double)ei[j]);   // i0 + 1 < 2 HS < n
        }
        return;
    }
    const double ycarry = lowpass_pairs_store<float, NT, R0>(er, ei, P.a1, P.b0, smem, y, 2 * HS, tid);
    F2_PSTAMP(6);
#pragma unroll
    for (int j = 0; j < R0; ++j) {
        const int i1 = 2 * (tid + j * NB0 + HS);
        const cpx<float> p = i1 < n ? *reinterpret_cast<const cpx<float>*>(y + i1) : cpx<float>{0.f, 0.f};
        er[j] = p.re;
        ei[j] = i1 + 1 < n ? p.im : 0.f;
    }
    __syncthreads();   // the scan's last LDS reads precede the next segment's writes
    lowpass_pairs_store<float, NT, R0>(er, ei, P.a1, P.b0, smem, y + 2 * HS, n - 2 * HS, tid, ycarry, *e_mid_p);
#ifdef F2_STAMPS
    F2_PSTAMP(7);
    if (tid == 0 && P.stamps)
        for (int k = 0; k < 8; ++k) P.stamps[(size_t)blockIdx.x * 8 + k] = pst[k];
#endif
}

template <int LOG2S>
int ensure_pair_tables(f2_ctx* ctx, f2_scratch& slot) {
    constexpr int HS = Geo<LOG2S>::HS;
    if (slot.ptr) return F2_OK;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<cpx<float>> host((size_t)HS + HS / 2);
    for (int m = 0; m < HS; ++m) {
        const long double ang = pi * (long double)m / (long double)HS;
        host[(size_t)m] = {(float)cosl(ang), (float)(-sinl(ang))};
    }
    for (int j = 0; j < HS / 2; ++j) {
        const long double ang = pi * (long double)(2 * j + 1) / (long double)(2 * HS);
        host[(size_t)HS + j] = {(float)cosl(ang), (float)(-sinl(ang))};
    }
    F2_TRY(f2_reserve(ctx, slot, sizeof(cpx<float>) * host.size()));
    F2_TRY(f2_upload_async(ctx, slot.ptr, host.data(), sizeof(cpx<float>) * host.size()));
    return F2_OK;
}

template <int LOG2S>
int launch_pair(f2_ctx* ctx, const double* d_gfb, double* d_env, const int64_t* d_offsets, const int64_t* h_offsets,
                const int* utts, int nutt, int C, int lpf, double b0, double a1, const float* d_x32,
                const int64_t* d_x32_off, const int64_t* h_x32_off, const int* d_uflag) {
    constexpr int HS = Geo<LOG2S>::HS, NT = Geo<LOG2S>::NT;
    f2_scratch& tables = ctx->tw_pair[LOG2S - 13];
    F2_TRY(ensure_pair_tables<LOG2S>(ctx, tables));
    F2_TRY(ensure_twiddles<float>(ctx, LOG2S, ctx->tw[0][LOG2S]));
    f2_scratch& list = ctx->pair_list[LOG2S - 13];
    std::vector<int>& list_host = ctx->pair_list_host[LOG2S - 13];
    if (!list.ptr || list_host.size() != (size_t)nutt || memcmp(list_host.data(), utts, sizeof(int) * (size_t)nutt) != 0) {
        // (same batch shape as the previous call: the device copy is still valid and the call only enqueues)
        list_host.clear();
        F2_TRY(f2_reserve(ctx, list, sizeof(int) * (size_t)nutt));
        F2_TRY(f2_upload_async(ctx, list.ptr, utts, sizeof(int) * (size_t)nutt));
        list_host.assign(utts, utts + nutt);
    }
    PairParams P;
    P.gfb = d_gfb;
    P.env = d_env;
    P.offsets = d_offsets;
    P.ulist = nullptr;
    P.C = C;
    P.lpf = lpf;
    P.b0 = b0;
    P.a1 = a1;
    P.x32 = d_x32;
    P.x32_off = d_x32_off;
    P.tx = (const cpx<float>*)tables.ptr;
    P.vo = P.tx + HS;
    P.ulist = (const int*)list.ptr;
    P.uflag = d_uflag;
    P.stamps = nullptr;
#ifdef F2_STAMPS
    static unsigned long long* d_stamps = nullptr;
    const size_t nstamp = (size_t)nutt * C * 8;
    if (!d_stamps) F2_HIP(ctx, hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 8 * 128 * 2048));
    if (nstamp <= (size_t)8 * 128 * 2048) P.stamps = d_stamps;
#endif
    const dim3 grid((unsigned)((size_t)nutt * C)), block(NT);
    const cpx<float>* tw = (const cpx<float>*)ctx->tw[0][LOG2S].ptr;
    F2_TRY(f2_prof_begin(ctx, F2_K_ENVELOPE));
    if (d_x32)
        hipLaunchKernelGGL((k_envelope_pair<LOG2S, float>), grid, block, 0, ctx->stream, P, tw);
    else
        hipLaunchKernelGGL((k_envelope_pair<LOG2S, double>), grid, block, 0, ctx->stream, P, tw);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_ENVELOPE));
#ifdef F2_STAMPS
    if (P.stamps) {
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<unsigned long long> h(nstamp);
        F2_HIP(ctx, hipMemcpy(h.data(), d_stamps, sizeof(unsigned long long) * nstamp, hipMemcpyDeviceToHost));
        double acc[8] = {0};
        const size_t rows = (size_t)nutt * C;
        for (size_t r = 0; r < rows; ++r)
            for (int k = 1; k < 8; ++k) acc[k] += (double)(h[r * 8 + k] - h[r * 8 + k - 1]);
        static const char* names[8] = {"", "load+radix2", "even sub-row", "park E/fetch b", "odd sub-row", "last stage+magnitude",
                                       "barrier+lpf lower", "lpf upper"};
        fprintf(stderr, "[pair stamps] mean 10 ns ticks per workgroup:");
        for (int k = 1; k < 8; ++k) fprintf(stderr, " %s=%.0f", names[k], acc[k] / rows);
        fprintf(stderr, "\n");
    }
#endif
    return F2_OK;
}

}  // namespace

// Rows of 2^15 complex points (32768 < n <= 65536), float transforms, input not aliased with the output rows. (The same
// kernel with 8192-point sub-rows, two workgroups per CU, was measured for 16385..32768-sample rows: 12.2 ms against the
// single-row 1024-thread plan's 8.0 ms for 16 M samples - the parking and the extra stages cost more than the second
// workgroup buys - so those rows stay with f2_envelope.hip.)
bool f2_envelope_pair_supports(int log2h, int precision) { return precision == F2_FFT_F32 && log2h == 15; }

// All utterances `utts` have the same transform size 2^log2h complex points.
int f2_launch_envelope_pair(f2_ctx* ctx, const double* d_gfb, double* d_env, const int64_t* d_offsets,
                            const int64_t* h_offsets, const int* utts, int nutt, int log2h, int C, int lpf, double b0,
                            double a1, const float* d_x32, const int64_t* d_x32_off, const int64_t* h_x32_off,
                            const int* d_uflag) {
    if (nutt <= 0) return F2_OK;
    F2_CHECK(ctx, d_x32 || d_gfb != d_env, F2_ERR_INVALID, "the two-sub-row envelope path cannot run in place");
    F2_CHECK(ctx, f2_envelope_pair_supports(log2h, F2_FFT_F32), F2_ERR_INVALID, "unsupported size 2^%d", log2h);
    return launch_pair<14>(ctx, d_gfb, d_env, d_offsets, h_offsets, utts, nutt, C, lpf, b0, a1, d_x32, d_x32_off, h_x32_off,
                           d_uflag);
}
