// K2 for rows that do not fit in LDS (32768 < n <= 262144 samples, float transforms): the same mathematics as
// f2_envelope.hip (reference: scripts/processing/EnvelopeExtraction.py:20-67) with the H-point transforms factored
// H = H1 x H2, H2 = 4096 points LDS-resident, H1 = 8 / 16 / 32 in registers ("four-step" FFT). Three passes over HBM
// instead of one launch per radix-16 stage:
//
//   KA  z[m] = x[2m] + i x[2m+1], m = n2 + H2 n1.  Thread n2 loads its H1 points (coalesced across n2), runs the
//       H1-point DFT over n1 in registers, multiplies by exp(-2 pi i n2 k1 / H) and writes A[k1][n2]:
//         Z[k1 + H1 k2] = sum_n2 e^{-2 pi i n2 k2 / H2} A[k1][n2]
//   KB  one workgroup = the sub-transforms k1 and H1 - k1 of one row, one per 256-thread half, both in LDS:
//       forward 4096-point FFTs (fft_all), then the Hilbert pair step - Z[k] meets Z[H-k], and
//       H - (k1 + H1 k2) = (H1 - k1) + H1 (H2 - 1 - k2) lives in the other half's LDS (k1 = 0: H1 (H2 - k2), own) -
//       W~[k] = conj(W[k]) / H,  W[k] = i sin(t_k) Z[k] + cos(t_k) conj(Z[H-k]),  t_k = pi k / H,  W[0] = 0,
//       formed in registers with e^{i t_k} = e^{i pi k1 / H} e^{i pi k2 / H2} (two small tables), then the second
//       4096-point FFTs (registers -> registers) give C[k1][m2] = sum_k2 W~[k1 + H1 k2] e^{-2 pi i k2 m2 / H2}
//   KC  thread m2 loads C[k1][m2] for all k1, multiplies by exp(-2 pi i k1 m2 / H), runs the H1-point DFT over k1:
//         w~[m2 + H2 m1] (only its squares are used), env[2m] = sqrt(x[2m]^2 + Re^2), env[2m+1] = sqrt(x[2m+1]^2 + Im^2)
//       written as float64 (in place over x when the caller transforms in place: a thread reads exactly the x it
//       overwrites)
//   KL  optional low-pass (KC then leaves the envelope as float pairs in the scratch row instead of writing float64),
//       one workgroup per row, 16384-sample segments through lowpass_pairs_store with the filter state carried from
//       segment to segment (coalesced 8-byte loads, 16-byte stores)
//
// A and C share one scratch array of H complex floats per row; utterances are processed in groups that keep it
// under SCRATCH_CAP bytes. Algorithmic bytes per sample-channel: 8+4 (KA) + 4+4 (KB) + 4+8+4 (KC) + 4+8 (KL), KC 4+8+8 without low-pass; the reads
// of x drop to 4 bytes when the filterbank handed its rows over as float32 (x32: compact rows in a scratch buffer - not
// in the output slots, where KC's float64 stores would overwrite samples other workgroups have not read yet).
#include <algorithm>

#include "f2_fft_lds.h"

using namespace f2fft;

namespace {

constexpr int LOG2H2 = 12;
constexpr int H2 = 1 << LOG2H2;
constexpr int NTH = 256;                 // threads per sub-transform (threads_for<float, 12>())
constexpr size_t SCRATCH_CAP = size_t(3) << 30;

struct SplitParams {
    const double* gfb;
    double* env;
    const int64_t* offsets;
    const int* ulist;          // utterances of this launch (device)
    const int* uflag;          // per utterance of the batch, or NULL: rows of utterances whose flag is 0 are left alone
    int C;
    int lpf;
    double b0, a1;
    cpx<float>* scratch;       // [row][H]
    const cpx<float>* twa;     // [H1][H2]: exp(-2 pi i k1 n2 / H)
    const cpx<float>* t1;      // [H1]: (cos, sin)(pi k1 / H)
    const cpx<float>* t2;      // [H2]: (cos, sin)(pi k2 / H2)
    const cpx<float>* tw12;    // tables of the 4096-point transform (ensure_twiddles<float>(12))
    const float* x32;          // float32 hand-off of the filterbank: compact (C, n) rows at x32 + x32_off[b], or NULL
    const int64_t* x32_off;
};

struct Row {
    const double* x;     // float64 input row (when xf is NULL)
    const float* xf;     // float32 input row, or NULL
    double* y;
    int n;
};
// utterances the spectral kernel has served (flag 0) are skipped by every pass: workgroup-uniform
__device__ __forceinline__ bool row_skipped(const SplitParams& P, int r) {
    return P.uflag && P.uflag[P.ulist[r / P.C]] == 0;
}
__device__ __forceinline__ Row row_of(const SplitParams& P, int r) {
    const int u = r / P.C, c = r - u * P.C;
    const int b = P.ulist[u];
    const int64_t off = P.offsets[b];
    const int n = (int)(P.offsets[b + 1] - off);
    const size_t row = (size_t)P.C * (size_t)off + (size_t)c * (size_t)n;
    const float* xf = P.x32 ? P.x32 + P.x32_off[b] + (size_t)c * (size_t)n : nullptr;
    return {P.gfb + row, xf, P.env + row, n};
}
// samples i0, i0 + 1 of the row as floats (0 beyond the end): one wide load from a clamped address
__device__ __forceinline__ cpx<float> load_pair(const Row& rw, int i0) {
    cpx<float> r;
    if (rw.xf) row_pair<true>(rw.xf, rw.n, i0, r.re, r.im);
    else row_pair<true>(rw.x, rw.n, i0, r.re, r.im);
    return r;
}

template <int H1>
__global__ __launch_bounds__(256) void k_split_first(SplitParams P) {
    const int r = blockIdx.x, n2 = blockIdx.y * 256 + threadIdx.x;
    if (row_skipped(P, r)) return;
    const Row rw = row_of(P, r);
    cpx<float> v[H1];
#pragma unroll
    for (int n1 = 0; n1 < H1; ++n1) {
        v[n1] = load_pair(rw, 2 * (n2 + H2 * n1));
    }
    dft<H1>(v);   // X[k1] in v[brev<H1>(k1)]
    cpx<float>* A = P.scratch + (size_t)r * (H1 * H2);
#pragma unroll
    for (int k1 = 0; k1 < H1; ++k1) {
        const cpx<float> t = k1 == 0 ? v[0] : cmul(v[brev<H1>(k1)], P.twa[k1 * H2 + n2]);
        A[k1 * H2 + n2] = t;
    }
}

template <int H1>
__global__ __launch_bounds__(2 * NTH, 4) void k_split_mid(SplitParams P) {
    constexpr int CS = cpad_size(H2);
    constexpr int PT = plan_points_per_thread(LOG2H2, NTH);   // 16
    constexpr int R0 = 1 << plan_bits(LOG2H2, 0);             // 16
    constexpr int NB0 = H2 / R0;                              // 256
    static_assert(PT == R0 && NB0 == NTH, "one radix-16 butterfly per thread in the first and last pass");
    constexpr int TWL = plan_tw_lds_count(LOG2H2);
    constexpr bool T0R = derive_tw0<float, LOG2H2>();
    __shared__ __attribute__((aligned(16))) cpx<float> smem[2 * CS];
    __shared__ __attribute__((aligned(16))) cpx<float> twl[TWL > 0 ? TWL : 1];

    const int tid = threadIdx.x, half = tid >> 8, t = tid & (NTH - 1);
    const int r = blockIdx.x, p = blockIdx.y;                // pair p = (p, H1 - p), 0 <= p <= H1/2
    if (row_skipped(P, r)) return;
    const bool self = p == 0 || 2 * p == H1;                 // the sub-transform pairs with itself
    const int ka = half == 0 ? p : (H1 - p) % H1;
    cpx<float>* own = smem + half * CS;
    const cpx<float>* partner = self ? own : smem + (half ^ 1) * CS;
    for (int i = tid; i < TWL; i += 2 * NTH) twl[i] = P.tw12[plan_tw_offset(LOG2H2, 1) + i];

    cpx<float>* A = P.scratch + (size_t)r * (H1 * H2) + (size_t)ka * H2;
    cpx<float> v[PT];
#pragma unroll
    for (int j = 0; j < R0; ++j) v[j] = A[t + j * NB0];
    fft_all<float, LOG2H2, false, PT, NTH, T0R>(own, P.tw12, twl, t, v);   // Z[ka + H1 k2] at own[cpad(k2)]

    // Hilbert pair step in registers; all reads of both halves' spectra precede the next transform's LDS writes
    const cpx<float> r1 = P.t1[ka];
    const float sc = 1.0f / (float)(H1 * H2);
    // (v is dead after the forward transform - its outputs went to LDS - so W~ is formed straight into it, four points at
    // a time to keep the loads in flight without holding all 48 operands)
#pragma unroll
    for (int j0 = 0; j0 < R0; j0 += 4) {
        cpx<float> zk[4], zm[4], r2[4];
        int tp = t;
        asm volatile("" : "+v"(tp));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k2 = tp + (j0 + q) * NB0;
            const int km = ka == 0 ? ((H2 - k2) & (H2 - 1)) : H2 - 1 - k2;
            zk[q] = own[cpad(k2)];
            zm[q] = partner[cpad(km)];
            r2[q] = P.t2[k2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float cs = (r1.re * r2[q].re - r1.im * r2[q].im) * sc;   // cos(t_k) / H
            const float sn = (r1.im * r2[q].re + r1.re * r2[q].im) * sc;   // sin(t_k) / H
            const bool dc = ka == 0 && t + (j0 + q) * NB0 == 0;
            v[j0 + q] = {dc ? 0.f : -sn * zk[q].im + cs * zm[q].re, dc ? 0.f : -(sn * zk[q].re - cs * zm[q].im)};
        }
    }
    __syncthreads();   // all reads of both halves' spectra precede the next transform's LDS writes
    fft_regs_to_regs<float, LOG2H2, PT, NTH, T0R>(own, P.tw12, twl, t, v);   // point t + j*NB0 in v[brev<R0>(j)]
    if (half == 0 || !self) {
        int ts = t;
        asm volatile("" : "+v"(ts));   // (the 16 addresses are formed again here, not held through both transforms)
#pragma unroll
        for (int j = 0; j < R0; ++j) A[ts + j * NB0] = v[brev<R0>(j)];
    }
}

template <int H1>
__global__ __launch_bounds__(256) void k_split_last(SplitParams P) {
    const int r = blockIdx.x, m2 = blockIdx.y * 256 + threadIdx.x;
    if (row_skipped(P, r)) return;
    const Row rw = row_of(P, r);
    const cpx<float>* Cm = P.scratch + (size_t)r * (H1 * H2);
    cpx<float>* Ce = P.scratch + (size_t)r * (H1 * H2);
    cpx<float> v[H1];
#pragma unroll
    for (int k1 = 0; k1 < H1; ++k1) {
        const cpx<float> c = Cm[k1 * H2 + m2];
        v[k1] = k1 == 0 ? c : cmul(c, P.twa[k1 * H2 + m2]);
    }
    dft<H1>(v);   // w~[m2 + H2 m1] in v[brev<H1>(m1)]
#pragma unroll
    for (int m1 = 0; m1 < H1; ++m1) {
        const int i0 = 2 * (m2 + H2 * m1);
        const cpx<float> w = v[brev<H1>(m1)];
        const cpx<float> xx = load_pair(rw, i0);   // (with float64 rows in place: exactly the samples overwritten below)
        const float f0 = fsqrt(xx.re * xx.re + w.re * w.re), f1 = fsqrt(xx.im * xx.im + w.im * w.im);
        if (P.lpf) {
            // the low-pass follows: leave the envelope as float pairs in the scratch row, in time order - position
            // m2 + H2 m1 is one of the positions this thread has just read, so the row is rewritten in place
            Ce[m1 * H2 + m2] = {f0, f1};
            continue;
        }
        store_row_pair(rw.y, rw.n, i0, (double)f0, (double)f1);
    }
}

// first-order low-pass of the float envelope pairs k_split_last left in the scratch row -> float64 output row,
// 2*LNT*LNB = 16384 samples per step
constexpr int LNT = 512, LNB = 16;
template <int H1>
__global__ __launch_bounds__(LNT, 2) void k_split_lowpass(SplitParams P) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[lowpass_lds_bytes<float, LNT, LNB>()];
    __shared__ float e_last;
    const int tid = threadIdx.x;
    if (row_skipped(P, blockIdx.x)) return;
    const Row rw = row_of(P, blockIdx.x);
    double* y = rw.y;
    const int n = rw.n;
    const cpx<float>* __restrict__ e = P.scratch + (size_t)blockIdx.x * (H1 * H2);   // envelope pairs left by k_split_last
    double ycarry = 0.0;
    float ecarry = 0.f;
    for (int base = 0; base < n; base += 2 * LNT * LNB) {
        const int left = n - base;
        float er[LNB], ei[LNB];
#pragma unroll
        for (int jj = 0; jj < LNB; ++jj) {
            const int i0 = 2 * (tid + LNT * jj);
            const cpx<float> p = e[min((base + i0) >> 1, H1 * H2 - 1)];
            er[jj] = i0 < left ? p.re : 0.f;
            ei[jj] = i0 + 1 < left ? p.im : 0.f;
        }
        if (tid == LNT - 1) e_last = ei[LNB - 1];   // e[-1] of the next segment
        __syncthreads();   // every load of this segment precedes every store; smem and e_last are settled
        const float e_next = e_last;
        ycarry = lowpass_pairs_store<float, LNT, LNB>(er, ei, P.a1, P.b0, smem, y + base, left, tid, ycarry, ecarry);
        __syncthreads();   // the scan's last LDS reads precede the next segment's writes (smem, e_last)
        ecarry = e_next;
    }
}

int ensure_split_tables(f2_ctx* ctx, int log2h, f2_scratch& slot) {
    if (slot.ptr) return F2_OK;
    const int H1 = 1 << (log2h - LOG2H2);
    const int64_t H = int64_t(1) << log2h;
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<cpx<float>> host((size_t)H + H1 + H2);
    for (int k1 = 0; k1 < H1; ++k1)
        for (int n2 = 0; n2 < H2; ++n2) {
            const long double ang = 2.0L * pi * (long double)((int64_t)k1 * n2 % H) / (long double)H;
            host[(size_t)k1 * H2 + n2] = {(float)cosl(ang), (float)(-sinl(ang))};
        }
    for (int k1 = 0; k1 < H1; ++k1) {
        const long double ang = pi * (long double)k1 / (long double)H;
        host[(size_t)H + k1] = {(float)cosl(ang), (float)sinl(ang)};
    }
    for (int k2 = 0; k2 < H2; ++k2) {
        const long double ang = pi * (long double)k2 / (long double)H2;
        host[(size_t)H + H1 + k2] = {(float)cosl(ang), (float)sinl(ang)};
    }
    F2_TRY(f2_reserve(ctx, slot, sizeof(cpx<float>) * host.size()));
    F2_TRY(f2_upload_async(ctx, slot.ptr, host.data(), sizeof(cpx<float>) * host.size()));
    return F2_OK;
}

template <int H1>
int launch_group(f2_ctx* ctx, const SplitParams& P, int nutt) {
    const int nrows = nutt * P.C;
    hipLaunchKernelGGL(k_split_first<H1>, dim3((unsigned)nrows, H2 / 256), dim3(256), 0, ctx->stream, P);
    F2_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_split_mid<H1>, dim3((unsigned)nrows, H1 / 2 + 1), dim3(2 * NTH), 0, ctx->stream, P);
    F2_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_split_last<H1>, dim3((unsigned)nrows, H2 / 256), dim3(256), 0, ctx->stream, P);
    F2_HIP(ctx, hipGetLastError());
    if (P.lpf) {
        hipLaunchKernelGGL(k_split_lowpass<H1>, dim3((unsigned)nrows), dim3(LNT), 0, ctx->stream, P);
        F2_HIP(ctx, hipGetLastError());
    }
    return F2_OK;
}

}  // namespace

bool f2_envelope_split_supports(int log2h, int precision) {
    return precision == F2_FFT_F32 && log2h > 14 && log2h - LOG2H2 <= 5;
}

// All utterances `utts` (indices into the batch) have the same transform size 2^log2h complex points.
int f2_launch_envelope_split(f2_ctx* ctx, const double* d_gfb, double* d_env, const int64_t* d_offsets, const int* utts,
                             int nutt, int log2h, int C, int lpf, double b0, double a1, const float* d_x32,
                             const int64_t* d_x32_off, const int* d_uflag) {
    if (nutt <= 0) return F2_OK;
    F2_CHECK(ctx, f2_envelope_split_supports(log2h, F2_FFT_F32), F2_ERR_INVALID, "unsupported split size 2^%d", log2h);
    const int H1 = 1 << (log2h - LOG2H2);
    const size_t H = size_t(1) << log2h;
    F2_TRY(ensure_split_tables(ctx, log2h, ctx->tw_split[log2h]));
    F2_TRY(ensure_twiddles<float>(ctx, LOG2H2, ctx->tw[0][LOG2H2]));
    const size_t per_utt = sizeof(cpx<float>) * H * (size_t)C;
    const int per_group = (int)std::max<size_t>(1, std::min<size_t>((size_t)nutt, SCRATCH_CAP / per_utt));
    F2_TRY(f2_reserve(ctx, ctx->work, per_utt * (size_t)per_group));
    F2_TRY(f2_reserve(ctx, ctx->work3, sizeof(int) * (size_t)nutt));
    F2_TRY(f2_upload_async(ctx, ctx->work3.ptr, utts, sizeof(int) * (size_t)nutt));
    SplitParams P;
    P.gfb = d_gfb;
    P.env = d_env;
    P.offsets = d_offsets;
    P.C = C;
    P.lpf = lpf;
    P.b0 = b0;
    P.a1 = a1;
    P.scratch = (cpx<float>*)ctx->work.ptr;
    P.twa = (const cpx<float>*)ctx->tw_split[log2h].ptr;
    P.t1 = P.twa + H;
    P.t2 = P.t1 + H1;
    P.tw12 = (const cpx<float>*)ctx->tw[0][LOG2H2].ptr;
    P.x32 = d_x32;
    P.x32_off = d_x32_off;
    P.uflag = d_uflag;
    for (int done = 0; done < nutt; done += per_group) {
        const int g = std::min(per_group, nutt - done);
        P.ulist = (const int*)ctx->work3.ptr + done;
        F2_TRY(f2_prof_begin(ctx, F2_K_ENVELOPE));
        int rc;
        switch (H1) {
            case 8: rc = launch_group<8>(ctx, P, g); break;
            case 16: rc = launch_group<16>(ctx, P, g); break;
            default: rc = launch_group<32>(ctx, P, g); break;
        }
        if (rc != F2_OK) return rc;
        F2_TRY(f2_prof_end(ctx, F2_K_ENVELOPE));
    }
    return F2_OK;
}
