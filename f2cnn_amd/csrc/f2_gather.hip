// K3 -- window gather + per-window log min-max normalisation.
// Reference: scripts/processing/InputGenerator.py:73-80 (gather at given centres, cast to float32 at :83),
// scripts/CNN/Evaluating.py:76-80 (every-sample windows) and scripts/CNN/Training.py:13-28 (normalizeInput).
//
//   out[e, k, c] = env[c, centre_e + step*(k - radius)]           k < 2*radius+1, c < C
//   normalize:     (ln v - ln min)/(ln max - ln min) over the whole window, in float64, then float32;
//                  all-equal window -> zeros; any value <= 0 -> error flag (the reference raises ValueError)
//
// One 256-thread workgroup per window. The window (R*C float64 values, 11 KiB for 11x128) is gathered
// into LDS once, reduced for min/max there, and written out with c fastest, i.e. fully coalesced float32
// rows. In `cnn eval` mode consecutive windows read consecutive samples of the same envelope rows, so
// the strided gather is served by L2.
#include "f2_internal.h"

namespace {

constexpr int GT = 256;

__global__ __launch_bounds__(GT) void k_gather_windows(const double* __restrict__ env, int C, int64_t N,
                                                       const int64_t* __restrict__ centers, int64_t first_center,
                                                       int radius, int step, int normalize,
                                                       float* __restrict__ out, int* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* win = reinterpret_cast<double*>(smem_raw);
    __shared__ double red_min[GT / 64], red_max[GT / 64];

    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;
    const int R = 2 * radius + 1;
    const int total = R * C;
    const int64_t centre = centers ? centers[e] : first_center + e;

    double mn = INFINITY, mx = -INFINITY;
    for (int idx = tid; idx < total; idx += GT) {
        const int k = idx / C, c = idx - k * C;
        const double v = env[(size_t)c * (size_t)N + (size_t)(centre + (int64_t)step * (k - radius))];
        win[idx] = v;
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    }
    float* o = out + (size_t)e * (size_t)total;
    if (!normalize) {
        __syncthreads();
        for (int idx = tid; idx < total; idx += GT) o[idx] = (float)win[idx];
        return;
    }
    for (int d = 32; d > 0; d >>= 1) {
        mn = fmin(mn, __shfl_xor(mn, d));
        mx = fmax(mx, __shfl_xor(mx, d));
    }
    if ((tid & 63) == 0) {
        red_min[tid >> 6] = mn;
        red_max[tid >> 6] = mx;
    }
    __syncthreads();
    mn = red_min[0];
    mx = red_max[0];
    for (int w = 1; w < GT / 64; ++w) {
        mn = fmin(mn, red_min[w]);
        mx = fmax(mx, red_max[w]);
    }
    if (!(mn > 0.0)) {   // also catches NaN
        if (tid == 0) atomicOr(flag, 1);
        for (int idx = tid; idx < total; idx += GT) o[idx] = 0.f;
        return;
    }
    if (mn == mx) {
        for (int idx = tid; idx < total; idx += GT) o[idx] = 0.f;
        return;
    }
    const double lmn = log(mn);
    const double range = log(mx) - lmn;
    for (int idx = tid; idx < total; idx += GT) o[idx] = (float)((log(win[idx]) - lmn) / range);
}

// ---- every-sample windows (`cnn eval`: centres first_center + e, normalised): two passes over coalesced data ----
// A sample of the envelope matrix appears in 2 * radius + 1 windows per channel; the per-window kernel above reads it with
// an 8-byte strided access and takes its logarithm (float64) each time. Here:
//   k_log_columns:   L[c][t] = ln env[c][t] once per sample (coalesced in t), with the minimum / maximum over each group of 16
//                    channels - a window's minimum is the minimum of R x groups of those;
//   k_window_stats:  (ln min, range) of every window from those partial minima, once;
//   k_eval_windows:  a workgroup = WB consecutive windows x one tap k; it reads, per channel, the WB consecutive values
//                    L[c][centre_0 + w + step (k - radius)] (one 256-byte run instead of 32 strided reads), normalises with
//                    the window's (ln min, range) and turns the (window, channel) tile through LDS into 512-byte output rows.
// Same arithmetic as k_gather_windows - ln v, ln min, (ln v - ln min) / range in float64, then float32 - so the results are
// bit-identical to it (tests/test_gpu_windows_cnn.py compares the two).
constexpr int WB = 32;      // consecutive windows per workgroup
constexpr int LCH = 16;     // channels per thread of k_log_columns (its grid's second dimension walks the channel groups)

__global__ __launch_bounds__(256) void k_log_columns(const double* __restrict__ env, int C, int64_t N, int64_t t0, int64_t span,
                                                     double* __restrict__ L, double* __restrict__ pmin,
                                                     double* __restrict__ pmax) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= span) return;
    const int64_t t = t0 + j;
    const int c0 = blockIdx.y * LCH, c1 = min(C, c0 + LCH);
    double mn = INFINITY, mx = -INFINITY;
    if (t >= 0 && t < N) {
        if (c1 - c0 == LCH) {
            // (all sixteen loads in flight before the first logarithm: the dependent load -> log -> store chain per channel was
            // sixteen memory latencies long)
            double v[LCH];
#pragma unroll
            for (int u = 0; u < LCH; ++u) v[u] = env[(size_t)(c0 + u) * (size_t)N + (size_t)t];
#pragma unroll
            for (int u = 0; u < LCH; ++u) {
                L[(size_t)(c0 + u) * (size_t)span + (size_t)j] = log(v[u]);
                mn = fmin(mn, v[u]);
                mx = fmax(mx, v[u]);
            }
        } else {
            for (int c = c0; c < c1; ++c) {
                const double v = env[(size_t)c * (size_t)N + (size_t)t];
                L[(size_t)c * (size_t)span + (size_t)j] = log(v);
                mn = fmin(mn, v);
                mx = fmax(mx, v);
            }
        }
    }
    pmin[(size_t)blockIdx.y * (size_t)span + (size_t)j] = mn;     // minimum / maximum over this group's channels
    pmax[(size_t)blockIdx.y * (size_t)span + (size_t)j] = mx;
}

// (ln min, ln max - ln min, flag) of every window, once (k_eval_windows used to form them again in each of its R workgroups per
// block of windows: R x groups loads per window and two float64 logarithms, eleven times): stats[4 e .. 4 e + 3]; flag 1 = all
// values equal or a non-positive value (rows of zeros; the error flag is raised here)
__global__ __launch_bounds__(256) void k_window_stats(const double* __restrict__ pmin, const double* __restrict__ pmax, int groups,
                                                      int64_t span, int64_t n_windows, int radius, int step,
                                                      double* __restrict__ stats, int* __restrict__ flag) {
    // 32 consecutive windows x 8 lanes per window: lane p of a window takes the (tap, group) pairs p, p + 8, ... (a single
    // thread walking all 88 serialises as many L2 latencies); for a fixed pair the 32 windows read one 256-byte run
    const int w = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int64_t e = (int64_t)blockIdx.x * 32 + w;
    const int R = 2 * radius + 1, pairs = R * groups;
    double mn = INFINITY, mx = -INFINITY;
    if (e < n_windows) {
#pragma unroll 4
        for (int pq = p; pq < pairs; pq += 8) {
            const int k2 = pq / groups, g = pq - k2 * groups;
            const size_t at = (size_t)g * (size_t)span + (size_t)(e + (int64_t)step * k2);
            mn = fmin(mn, pmin[at]);
            mx = fmax(mx, pmax[at]);
        }
    }
    __shared__ double smn[8][32], smx[8][32];
    smn[p][w] = mn;
    smx[p][w] = mx;
    __syncthreads();
    if (p != 0 || e >= n_windows) return;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        mn = fmin(mn, smn[q][w]);
        mx = fmax(mx, smx[q][w]);
    }
    double zero = 0.0;
    if (!(mn > 0.0)) {   // also catches NaN; the reference raises ValueError
        atomicOr(flag, 1);
        zero = 1.0;
    }
    if (mn == mx) zero = 1.0;
    const double lmn = log(mn), range = log(mx) - lmn;
    stats[4 * e] = lmn;
    stats[4 * e + 1] = range;
    stats[4 * e + 2] = zero;
    stats[4 * e + 3] = 1.0 / range;   // (correctly rounded: k_eval_windows divides by `range` through it)
}

// grid (blocks of WB windows, taps)
__global__ __launch_bounds__(256) void k_eval_windows(const double* __restrict__ L, const double* __restrict__ stats, int C,
                                                      int64_t span, int64_t n_windows, int radius, int step,
                                                      float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);            // [WB][C + 1]
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * WB;
    const int k = blockIdx.y;
    const int nw = (int)((n_windows - e0) < WB ? (n_windows - e0) : WB);
    const int R = 2 * radius + 1, CP = C + 1;
    // window e has its centre at span index e + step * radius (k_log_columns started `reach` samples before the first centre)
    const int w = tid & (WB - 1), cc = tid / WB;                 // 8 channel lanes x 32 windows
    const bool live = w < nw;
    const double* st = stats + 4 * (e0 + (live ? w : 0));
    const double lmn = st[0], range = st[1], rinv = st[3];
    const bool zero = st[2] != 0.0;
    const double* Lk = L + (e0 + w + (int64_t)step * k);
    // (eight loads in flight per thread; a / range, correctly rounded, in three float64 operations instead of the ~12 + v_rcp_f64
    // of a division: q0 = RN(a rinv) is within an ulp of the quotient, the residual a - q0 range is exact in one fma, and
    // RN(q0 + residual x rinv) is the rounded quotient - Markstein's correction step with rinv = RN(1 / range); bit-identical to
    // the per-window kernel's division in test_every_sample_windows_blocked_and_per_window_kernels_agree)
    constexpr int CL = 256 / WB, UN = 8;
    for (int c0 = cc; c0 < C; c0 += CL * UN) {
        double v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = (live && c0 + CL * u < C) ? Lk[(size_t)(c0 + CL * u) * (size_t)span] : lmn;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const double a = v[u] - lmn, q0 = a * rinv;
            const float o = (live && !zero) ? (float)fma(fma(-q0, range, a), rinv, q0) : 0.f;
            if (c0 + CL * u < C) tile[w * CP + c0 + CL * u] = o;
        }
    }
    __syncthreads();
    float* orow = out + ((size_t)e0 * R + k) * (size_t)C;
    if ((C & 3) == 0) {
        // 16 bytes per lane: a window's row of C floats is C / 4 consecutive lanes
        const int C4 = C >> 2;
        for (int idx = tid; idx < nw * C4; idx += 256) {
            const int ww = idx / C4, c = (idx - ww * C4) * 4;
            const float* t = tile + ww * CP + c;
            *reinterpret_cast<float4*>(orow + (size_t)ww * R * C + c) = make_float4(t[0], t[1], t[2], t[3]);
        }
    } else {
        for (int idx = tid; idx < nw * C; idx += 256) {
            const int ww = idx / C, c = idx - ww * C;
            orow[(size_t)ww * R * C + c] = tile[ww * CP + c];
        }
    }
}

}  // namespace

static int launch_eval_windows(f2_ctx* ctx, const double* d_env, int C, int64_t N, int64_t first_center, int64_t n_windows,
                               int radius, int step, float* d_out, int* d_flag) {
    const int64_t reach = (int64_t)step * radius;
    const int64_t span = n_windows + 2 * reach;
    const int groups = (C + LCH - 1) / LCH;
    F2_TRY(f2_reserve(ctx, ctx->gather_log, sizeof(double) * ((size_t)span * ((size_t)C + 2 * (size_t)groups) + 4 * (size_t)n_windows)));
    double* L = (double*)ctx->gather_log.ptr;
    double* pmin = L + (size_t)span * (size_t)C;
    double* pmax = pmin + (size_t)span * (size_t)groups;
    double* stats = pmax + (size_t)span * (size_t)groups;
    hipLaunchKernelGGL(k_log_columns, dim3((unsigned)((span + 255) / 256), (unsigned)groups), dim3(256), 0, ctx->stream, d_env, C, N,
                       first_center - reach, span, L, pmin, pmax);
    F2_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_window_stats, dim3((unsigned)((n_windows + 31) / 32)), dim3(256), 0, ctx->stream, (const double*)pmin,
                       (const double*)pmax, groups, span, n_windows, radius, step, stats, d_flag);
    F2_HIP(ctx, hipGetLastError());
    const size_t lds = sizeof(float) * WB * ((size_t)C + 1);
    hipLaunchKernelGGL(k_eval_windows, dim3((unsigned)((n_windows + WB - 1) / WB), (unsigned)(2 * radius + 1)), dim3(256), lds,
                       ctx->stream, (const double*)L, (const double*)stats, C, span, n_windows, radius, step, d_out);
    F2_HIP(ctx, hipGetLastError());
    return F2_OK;
}

int f2_launch_gather(f2_ctx* ctx, const double* d_env, int C, int64_t N, const int64_t* d_centers,
                     int64_t first_center, int64_t n_windows, int radius, int step, int normalize, float* d_out, int* d_flag) {
    if (n_windows <= 0) return F2_OK;
    const int R = 2 * radius + 1;
    const size_t lds = sizeof(double) * (size_t)R * (size_t)C;
    F2_CHECK(ctx, lds <= 150 * 1024, F2_ERR_UNSUPPORTED, "window of %d x %d values does not fit in LDS", R, C);
    F2_CHECK(ctx, n_windows < (int64_t(1) << 31), F2_ERR_UNSUPPORTED, "too many windows (%lld)", (long long)n_windows);
    if (lds > 64 * 1024)
        F2_HIP(ctx, hipFuncSetAttribute((const void*)k_gather_windows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    F2_TRY(f2_prof_begin(ctx, F2_K_GATHER));
    // every-sample normalised windows (`cnn eval`): the two-pass coalesced form; anything else one workgroup per window
    if (!d_centers && normalize && ctx->opt_gather_blocked && n_windows >= 4 * WB &&
        sizeof(float) * WB * ((size_t)C + 1) <= 64 * 1024) {
        F2_TRY(launch_eval_windows(ctx, d_env, C, N, first_center, n_windows, radius, step, d_out, d_flag));
        F2_TRY(f2_prof_end(ctx, F2_K_GATHER));
        return F2_OK;
    }
    hipLaunchKernelGGL(k_gather_windows, dim3((unsigned)n_windows), dim3(GT), lds, ctx->stream, d_env, C, N, d_centers,
                       first_center, radius, step, normalize, d_out, d_flag);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_GATHER));
    return F2_OK;
}
