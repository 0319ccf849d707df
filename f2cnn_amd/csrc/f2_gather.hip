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

}  // namespace

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
    hipLaunchKernelGGL(k_gather_windows, dim3((unsigned)n_windows), dim3(GT), lds, ctx->stream, d_env, C, N, d_centers,
                       first_center, radius, step, normalize, d_out, d_flag);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_GATHER));
    return F2_OK;
}
