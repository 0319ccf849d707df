// K1 -- ERB gammatone filterbank (reference: gammatone/filters.py:195-239 erb_filterbank).
//
// Work split: ONE WAVEFRONT = 64 channels of one utterance; lane = channel. The four cascaded
// second-order sections of a channel (shared poles B1,B2; zeros A11..A14) are advanced sample by
// sample with all eight float64 state words in registers: direct form II with three FMAs per section
// when the numerators have no z^-2 term (always, for make_erb_filters), else the transposed direct
// form II recurrence scipy.signal.lfilter evaluates. The input sample is the same for all 64 lanes, so a
// block of TB samples is staged once in LDS (fetched one block ahead) and read back as a broadcast.
//
// The (C,N) C-order output would make every lane store to its own row (stride N*8 bytes). Instead a
// TB-sample block of results is written to an LDS tile [64 lanes][TB+1] (pad 1 double: conflict-free
// ds_write_b64 column writes) and then streamed out row by row, so every global store instruction
// covers 64/TB full row segments of TB*8 contiguous bytes.
//
// Bound: the float64 FMA pipe (13 ops per sample-channel; 17 in the general form); HBM traffic is 8*C*N bytes
// written per utterance (4*C*N as the float32 hand-off to K2) + 2*N read.
#include "f2_internal.h"

namespace {

constexpr int TB = 32;                 // samples per LDS tile
constexpr int ROWS_PER_STORE = 64 / TB;
// Wavefronts per workgroup. The waves of a workgroup are independent (own utterance/channel group, own LDS tile, no
// workgroup barrier); they only share a workgroup so that the dispatcher places them one per SIMD of a CU. With
// one-wave workgroups the 2000 waves of the benchmark batch land unevenly (some SIMDs run three, others one) and the
// kernel takes as long as the fullest SIMD.
#ifndef F2_K1_WAVES_F32
#define F2_K1_WAVES_F32 4     // float32 hand-off tiles: 4 x 8.4 KB of LDS
#endif
#ifndef F2_K1_WAVES_F64
#define F2_K1_WAVES_F64 2     // float64 output tiles: 2 x 16.9 KB (the store-bound variant gains nothing beyond two)
#endif
template <typename OutT>
constexpr int waves_per_block() { return sizeof(OutT) == 4 ? F2_K1_WAVES_F32 : F2_K1_WAVES_F64; }

// LDS hand-over between lanes of ONE wave: the wave's DS operations execute in order, so only the compiler has to be
// kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// OutT = double: the (C,N) float64 matrix of the reference. OutT = float: the same row layout but each row's
// float32 samples sit at the START of that row's float64 slot (row r, sample t -> float index 2*r*N + t), the
// hand-off format to the float32-FFT envelope kernel, which then overwrites the slot with float64 envelopes.
// A2ZERO: the numerators have no z^-2 term (always true for make_erb_filters output): direct form II sections,
// 13 float64 ops per sample-channel; otherwise the general transposed-direct-form-II recurrences (17 ops).
template <typename WaveT, typename OutT, bool A2ZERO>
__global__ __launch_bounds__(64 * waves_per_block<OutT>()) void k_erb_filterbank(const WaveT* __restrict__ wave,
                                                       const int64_t* __restrict__ offsets,
                                                       const double* __restrict__ coefs, int C,
                                                       int groups, int units, double* __restrict__ out,
                                                       float* __restrict__ alt, const int64_t* __restrict__ alt_off) {
    constexpr int WPB = waves_per_block<OutT>();
    __shared__ OutT tiles[WPB][64][TB + 1];
    __shared__ double xss[WPB][TB];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int unit = blockIdx.x * WPB + wv;          // (utterance, group of 64 channels)
    if (unit >= units) return;                       // whole wave: the waves of a workgroup never meet at a barrier
    OutT (*tile)[TB + 1] = tiles[wv];
    double* xs = xss[wv];
    const int b = unit / groups;
    const int c0 = (unit % groups) * 64;
    const int64_t off = offsets[b];
    const int64_t N = offsets[b + 1] - off;
    if (N <= 0) return;

    const int c = min(c0 + lane, C - 1);  // idle lanes shadow the last channel; their rows are never stored
    const double* k = coefs + (size_t)c * 10;
    // lfilter normalises by a[0]; make_erb_filters always emits B0 == 1 and A2 == 0, kept general here
    const double rB0 = 1.0 / k[6];
    const double inv_gain = 1.0 / k[9];
    const double b0 = k[0] * rB0, b2 = k[5] * rB0, a1 = k[7] * rB0, a2 = k[8] * rB0;
    const double b11 = k[1] * rB0, b12 = k[2] * rB0, b13 = k[3] * rB0;
    // last section scaled by 1/gain: its output is the final sample
    const double b0g = b0 * inv_gain, b14g = k[4] * rB0 * inv_gain, b2g = b2 * inv_gain;
    // A2ZERO form: zero of section k at -ck, overall factor b0^4 / gain
    const double c1 = k[1] / k[0], c2 = k[2] / k[0], c3 = k[3] / k[0], c4 = k[4] / k[0];
    const double scale = (b0 * b0) * (b0 * b0) * inv_gain;

    double z10 = 0, z11 = 0, z20 = 0, z21 = 0, z30 = 0, z31 = 0, z40 = 0, z41 = 0;

    const WaveT* w = wave + off;
    // row pitch in OutT elements = ROWMUL * N: float rows sit at the start of their float64 slot, unless this utterance
    // hands its rows over in the compact scratch layout (alt_off[b] >= 0: (C, N) float rows at alt + alt_off[b])
    int ROWMUL = sizeof(double) / sizeof(OutT);
    char* obase = reinterpret_cast<char*>(out + (size_t)C * (size_t)off);
    if (sizeof(OutT) == 4 && alt_off) {
        const int64_t ao = alt_off[b];
        if (ao >= 0) {
            ROWMUL = 1;
            obase = reinterpret_cast<char*>(alt + ao);
        }
    }
    const int srow = lane / TB, scol = lane % TB;
    // byte offsets inside this utterance's block fit 32 bits up to 512 Mi sample-channels
    const bool fits32 = (uint64_t)C * (uint64_t)N * 8u < (uint64_t(1) << 32);
    const bool full_rows = c0 + 64 <= C;
    const uint32_t rstep = (uint32_t)(ROWS_PER_STORE * N * ROWMUL * sizeof(OutT));

    // The input block is fetched one block ahead: vmcnt retires in issue order (stores included), so a load
    // issued after a block's 32 row stores would make the wave wait for those stores every block.
    WaveT xnext = (lane < TB && lane < N) ? w[lane] : WaveT(0);
    for (int64_t t0 = 0; t0 < N; t0 += TB) {
        if (lane < TB) xs[lane] = (double)xnext;
        {
            const int64_t t = t0 + TB + lane;
            xnext = (lane < TB && t < N) ? w[t] : WaveT(0);
        }
        wave_sync();
        // The four sections of one sample form a dependent chain (y1 -> y2 -> y3 -> y4). The block is written
        // out skewed -- step s runs section k on sample s-k+1 -- so that every step holds four independent
        // recurrences and the in-order VALU always has a ready float64 FMA.
        double p1 = 0, p2 = 0, p3 = 0;
        if constexpr (A2ZERO) {
            // Numerators T + A1k z^-1 = T (1 + ck z^-1): each section in direct form II costs three FMAs,
            //   w = in - a1 w[-1] - a2 w[-2] ;  out = w + ck w[-1]
            // and the factor T^4 / gain is applied once to the last section's output: 13 float64 ops per
            // sample-channel. (z10/z11 ... hold w[-1]/w[-2] of sections 1..4 in this branch.)
#pragma unroll
            for (int s2 = 0; s2 < TB + 3; ++s2) {
                double n1 = 0, n2 = 0, n3 = 0;
                if (s2 < TB) {
                    const double wv = fma(-a1, z10, fma(-a2, z11, xs[s2]));
                    n1 = fma(c1, z10, wv);
                    z11 = z10;
                    z10 = wv;
                }
                if (s2 >= 1 && s2 - 1 < TB) {
                    const double wv = fma(-a1, z20, fma(-a2, z21, p1));
                    n2 = fma(c2, z20, wv);
                    z21 = z20;
                    z20 = wv;
                }
                if (s2 >= 2 && s2 - 2 < TB) {
                    const double wv = fma(-a1, z30, fma(-a2, z31, p2));
                    n3 = fma(c3, z30, wv);
                    z31 = z30;
                    z30 = wv;
                }
                if (s2 >= 3) {
                    const double wv = fma(-a1, z40, fma(-a2, z41, p3));
                    tile[lane][s2 - 3] = (OutT)(fma(c4, z40, wv) * scale);
                    z41 = z40;
                    z40 = wv;
                }
                p1 = n1;
                p2 = n2;
                p3 = n3;
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < TB + 3; ++s2) {
                double n1 = 0, n2 = 0, n3 = 0;
                if (s2 < TB) {
                    const double x = xs[s2];
                    // transposed direct form II: y = b0*x + z0 ; z0 = b1*x - a1*y + z1 ; z1 = b2*x - a2*y
                    const double y1 = fma(b0, x, z10);
                    z10 = fma(-a1, y1, fma(b11, x, z11));
                    z11 = fma(b2, x, -a2 * y1);
                    n1 = y1;
                }
                if (s2 >= 1 && s2 - 1 < TB) {
                    const double y2 = fma(b0, p1, z20);
                    z20 = fma(-a1, y2, fma(b12, p1, z21));
                    z21 = fma(b2, p1, -a2 * y2);
                    n2 = y2;
                }
                if (s2 >= 2 && s2 - 2 < TB) {
                    const double y3 = fma(b0, p2, z30);
                    z30 = fma(-a1, y3, fma(b13, p2, z31));
                    z31 = fma(b2, p2, -a2 * y3);
                    n3 = y3;
                }
                if (s2 >= 3) {
                    const double y4 = fma(b0g, p3, z40);
                    z40 = fma(-a1, y4, fma(b14g, p3, z41));
                    z41 = fma(b2g, p3, -a2 * y4);
                    tile[lane][s2 - 3] = (OutT)y4;
                }
                p1 = n1;
                p2 = n2;
                p3 = n3;
            }
        }
        wave_sync();
        if (fits32 && full_rows && t0 + TB <= N) {
            // whole tile inside the matrix: one 32-bit offset add per store, no checks
            uint32_t boff = (uint32_t)((((size_t)(c0 + srow) * (size_t)N) * ROWMUL + (size_t)(t0 + scol)) * sizeof(OutT));
            // all LDS reads first (distinct registers), then the stores: no load-use wait per row
            OutT vals[64 / ROWS_PER_STORE];
#pragma unroll
            for (int q = 0; q < 64 / ROWS_PER_STORE; ++q) vals[q] = tile[q * ROWS_PER_STORE + srow][scol];
#pragma unroll
            for (int q = 0; q < 64 / ROWS_PER_STORE; ++q) {
                *reinterpret_cast<OutT*>(obase + boff) = vals[q];
                boff += rstep;
            }
        } else {
            const int64_t t = t0 + scol;
            if (t < N) {
                OutT* o = reinterpret_cast<OutT*>(obase);
#pragma unroll 8
                for (int r = 0; r < 64; r += ROWS_PER_STORE) {
                    const int row = r + srow;
                    if (c0 + row < C) o[(size_t)(c0 + row) * (size_t)N * ROWMUL + (size_t)t] = tile[row][scol];
                }
            }
        }
        wave_sync();
    }
}

template <typename WaveT, typename OutT>
void launch_fb(hipStream_t st, int units, bool a2zero, const void* wave, const int64_t* offsets, const double* coefs, int C,
               int groups, double* out, float* alt = nullptr, const int64_t* alt_off = nullptr) {
    constexpr int WPB = waves_per_block<OutT>();
    const dim3 grid((unsigned)((units + WPB - 1) / WPB)), block(64 * WPB);
    if (a2zero)
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, true>), grid, block, 0, st, (const WaveT*)wave, offsets, coefs, C,
                           groups, units, out, alt, alt_off);
    else
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, false>), grid, block, 0, st, (const WaveT*)wave, offsets, coefs,
                           C, groups, units, out, alt, alt_off);
}

}  // namespace

int f2_launch_filterbank(f2_ctx* ctx, const void* d_wave, int wave_dtype, const int64_t* d_offsets,
                         const int64_t* h_offsets, const double* d_coefs, int B, int C, double* d_gfb,
                         const f2_handoff* handoff) {
    const bool f32_out = handoff && handoff->f32;
    float* alt = f32_out ? handoff->d_x32 : nullptr;
    const int64_t* alt_off = f32_out ? handoff->d_x32_off : nullptr;
    (void)h_offsets;
    const int groups = (C + 63) / 64;
    const int units = B * groups;
    // the coefficient rows of this call are mirrored on the host by f2_upload_coefs
    bool a2zero = ctx->coefs_host.size() == (size_t)C * 10;
    for (int c = 0; a2zero && c < C; ++c) a2zero = ctx->coefs_host[(size_t)c * 10 + 5] == 0.0;
    F2_TRY(f2_prof_begin(ctx, F2_K_FILTERBANK));
    if (wave_dtype == F2_WAVE_I16 && !f32_out)
        launch_fb<int16_t, double>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb);
    else if (wave_dtype == F2_WAVE_I16)
        launch_fb<int16_t, float>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb, alt, alt_off);
    else if (!f32_out)
        launch_fb<double, double>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb);
    else
        launch_fb<double, float>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb, alt, alt_off);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_FILTERBANK));
    return F2_OK;
}
