// K1 -- ERB gammatone filterbank (reference: gammatone/filters.py:195-239 erb_filterbank).
//
// Work split: ONE WAVEFRONT = 64 channels of one utterance; lane = channel. The four cascaded
// second-order sections of a channel (shared poles B1,B2; zeros A11..A14) are advanced sample by
// sample with all eight float64 state words in registers (transposed direct form II, the recurrence
// scipy.signal.lfilter evaluates). The input sample is the same for all 64 lanes, so a block of TB
// samples is staged once in LDS and read back as a broadcast.
//
// The (C,N) C-order output would make every lane store to its own row (stride N*8 bytes). Instead a
// TB-sample block of results is written to an LDS tile [64 lanes][TB+1] (pad 1 double: conflict-free
// ds_write_b64 column writes) and then streamed out row by row, so every global store instruction
// covers 64/TB full row segments of TB*8 contiguous bytes.
//
// Bound: HBM writes, 8*C*N bytes per utterance (+2*N read). Arithmetic: 17 f64 ops per sample-channel.
#include "f2_internal.h"

namespace {

constexpr int TB = 32;                 // samples per LDS tile
constexpr int ROWS_PER_STORE = 64 / TB;

template <typename WaveT>
__global__ __launch_bounds__(64) void k_erb_filterbank(const WaveT* __restrict__ wave,
                                                       const int64_t* __restrict__ offsets,
                                                       const double* __restrict__ coefs, int C,
                                                       int groups, double* __restrict__ out) {
    __shared__ double tile[64][TB + 1];
    __shared__ double xs[TB];

    const int lane = threadIdx.x;
    const int b = blockIdx.x / groups;
    const int c0 = (blockIdx.x % groups) * 64;
    const int64_t off = offsets[b];
    const int64_t N = offsets[b + 1] - off;
    if (N <= 0) return;

    const int c = min(c0 + lane, C - 1);  // idle lanes shadow the last channel; their rows are never stored
    const double* k = coefs + (size_t)c * 10;
    const double A0 = k[0], A11 = k[1], A12 = k[2], A13 = k[3], A14 = k[4];
    const double A2 = k[5], B0 = k[6];
    // lfilter normalises by a[0]; make_erb_filters always emits B0 == 1 and A2 == 0, kept general here
    const double rB0 = 1.0 / B0;
    const double b0 = A0 * rB0, b2 = A2 * rB0, a1 = k[7] * rB0, a2 = k[8] * rB0;
    const double b11 = A11 * rB0, b12 = A12 * rB0, b13 = A13 * rB0, b14 = A14 * rB0;
    const double inv_gain = 1.0 / k[9];

    double z10 = 0, z11 = 0, z20 = 0, z21 = 0, z30 = 0, z31 = 0, z40 = 0, z41 = 0;

    const WaveT* w = wave + off;
    double* o = out + (size_t)C * (size_t)off;
    const int srow = lane / TB, scol = lane % TB;

    for (int64_t t0 = 0; t0 < N; t0 += TB) {
        if (lane < TB) {
            const int64_t t = t0 + lane;
            xs[lane] = t < N ? (double)w[t] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const double x = xs[j];
            // section 1..4: y = b0*x + z0 ; z0 = b1*x - a1*y + z1 ; z1 = b2*x - a2*y
            const double y1 = fma(b0, x, z10);
            z10 = fma(-a1, y1, fma(b11, x, z11));
            z11 = fma(b2, x, -a2 * y1);
            const double y2 = fma(b0, y1, z20);
            z20 = fma(-a1, y2, fma(b12, y1, z21));
            z21 = fma(b2, y1, -a2 * y2);
            const double y3 = fma(b0, y2, z30);
            z30 = fma(-a1, y3, fma(b13, y2, z31));
            z31 = fma(b2, y2, -a2 * y3);
            const double y4 = fma(b0, y3, z40);
            z40 = fma(-a1, y4, fma(b14, y3, z41));
            z41 = fma(b2, y3, -a2 * y4);
            tile[lane][j] = y4 * inv_gain;
        }
        __syncthreads();
        const int64_t t = t0 + scol;
        if (t < N) {
#pragma unroll 8
            for (int r = 0; r < 64; r += ROWS_PER_STORE) {
                const int row = r + srow;
                if (c0 + row < C) o[(size_t)(c0 + row) * (size_t)N + (size_t)t] = tile[row][scol];
            }
        }
        __syncthreads();
    }
}

}  // namespace

int f2_launch_filterbank(f2_ctx* ctx, const void* d_wave, int wave_dtype, const int64_t* d_offsets,
                         const int64_t* h_offsets, const double* d_coefs, int B, int C, double* d_gfb) {
    (void)h_offsets;
    const int groups = (C + 63) / 64;
    const dim3 grid((unsigned)(B * groups)), block(64);
    F2_TRY(f2_prof_begin(ctx, F2_K_FILTERBANK));
    if (wave_dtype == F2_WAVE_I16)
        hipLaunchKernelGGL(k_erb_filterbank<int16_t>, grid, block, 0, ctx->stream, (const int16_t*)d_wave, d_offsets,
                           d_coefs, C, groups, d_gfb);
    else
        hipLaunchKernelGGL(k_erb_filterbank<double>, grid, block, 0, ctx->stream, (const double*)d_wave, d_offsets,
                           d_coefs, C, groups, d_gfb);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_FILTERBANK));
    return F2_OK;
}
