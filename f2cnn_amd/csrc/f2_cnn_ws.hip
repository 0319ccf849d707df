// K4, weight-stationary form of the split-bf16 convolutions (round 4) for the reference's window shape (10 or 11 rows:
// four pooled rows after conv2). Reference: architecture scripts/CNN/Training.py:93-114, predict scripts/CNN/Evaluating.py:84-87.
//
// Why. k_conv12_bf16x3 / k_conv34_bf16x3 (f2_cnn.hip) issue MFMAs 37 % of the time: every 32 x 32 x 16 product loads both of
// its operands again - the activations from LDS and the weights through the vector L1, 2 KB per wave and step, 85 B/clk/CU
// against the 64 B/clk the L1 delivers - and conv1 runs as 36 float32 FMAs per pixel and channel quad in front of the matrix loop,
// in phases that all waves of a workgroup go through together.
//
// Here every wave keeps the weights of its role in registers for the whole launch (9 taps x 32 input channels x 32 outputs x 2
// bf16 pieces = 144 VGPRs; 256 registers per wave, one 8-wave workgroup per CU, persistent over tiles), the roles of a workgroup
// form a pipeline with ONE barrier per tile, and a wave computes two output rows so that a patch row fetched from LDS serves
// tap row dy of one and dy - 1 of the other (48 ds_read_b128 per 108 MFMAs):
//
//   k_conv12_ws   tile = (window, 32 conv2 columns), all 8 pooled-input rows.
//                 waves 4-7 ("producers"): conv1 of tile t + 1 ON THE MATRIX CORES - out^T (32 channels x 32 pixels) = W1^T (32 x
//                 16: nine taps, the bias against a constant 1, zeros) x taps^T (16 x 32 pixels), three split-bf16 MFMAs per 32
//                 pixels; with the channels on the accumulator's row index a lane holds four consecutive channels of its pixel,
//                 so ReLU + split + two 8-byte LDS stores per quad write the conv2 input patch ([pixel][32 channels] bf16, one
//                 image per piece, 16-byte chunks XOR-swizzled) - no longer the oracle's fmaf chain bit for bit, held to the same
//                 2e-5 / referee rule.
//                 waves 0-3 ("consumers"): conv2 rows 2w, 2w + 1 of tile t from the other patch buffer, 2 x 2 max-pool in the lane
//                 (the two rows are the wave's two accumulators), bias + ReLU, output already split in two bf16 pieces
//                 ([window][4][W/2 - 1][hi 32 | lo 32]) so that conv3's staging is a copy.
//   k_conv34_ws   tile = (window, 30 conv4 columns). waves 0-3: conv3 (output tile nt, row pair) of tile t + 1 from patch A into
//                 patch B (64 channels per pixel, both pieces; accumulator rows = output channels as above); waves 4-7: first
//                 copy tile t + 2 from HBM into the free patch A, then conv4 (output tile nt, K half) of tile t from patch B;
//                 the K halves meet through LDS one iteration later, where the first half pools, adds the bias and stores.
//
// Layout of an MFMA operand (tools/ubench/mfma_bf16_layout.hip): lane (i, h) supplies k = 8 h .. 8 h + 7 of row / column i;
// accumulator register q of lane (i, h) is row (q & 3) + 8 (q >> 2) + 4 h, column i.
#include "f2_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int C1 = 32, C2 = 32, C3 = 64, C4 = 64;
constexpr int PW = 34;            // patch width: 32 output columns + 2
constexpr int T34 = 30;           // conv4 output columns per tile

// byte offset of 16-byte chunk `chunk` of pixel `pixel`; PIXB = bytes per pixel and piece (64: 32 channels, 128: 64 channels).
// Sixteen consecutive pixels put any one chunk on sixteen different 16-byte slots of the 256-byte bank row: ds_read_b128 of
// 32 consecutive pixels is conflict-free.
template <int PIXB>
__device__ __forceinline__ int pix_off(int pixel, int chunk) {
    if constexpr (PIXB == 64) return pixel * 64 + ((chunk ^ ((pixel >> 2) & 3)) << 4);
    else return pixel * 128 + ((chunk ^ ((pixel >> 1) & 7)) << 4);
}

__device__ __forceinline__ void split4(const float v0, const float v1, const float v2, const float v3, bf16x4& hi, bf16x4& lo) {
    const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1, h2 = (__bf16)v2, h3 = (__bf16)v3;
    hi = bf16x4{h0, h1, h2, h3};
    lo = bf16x4{(__bf16)(v0 - (float)h0), (__bf16)(v1 - (float)h1), (__bf16)(v2 - (float)h2), (__bf16)(v3 - (float)h3)};
}

// ReLU as ONE instruction (fmaxf canonicalises its operand first: two v_max_f32 per value)
__device__ __forceinline__ float relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }

// a value the compiler must keep in a register from here on (it cannot re-load or re-derive it inside the tile loop)
__device__ __forceinline__ void pin(bf16x8& v) { asm volatile("" : "+v"(v)); }

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// Two output rows x 32 columns x 32 output channels of a 3 x 3 convolution over 32 input channels (18 steps of 16): the
// patch rows r0 .. r0 + 3 are fetched once (hi and lo piece of (pixel, 16 channels): two ds_read_b128), row pr serving tap row
// dy = pr of output row 0 and dy = pr - 1 of output row 1. SWAP: the weights are the A operand (accumulator rows = output
// channels, columns = pixels), otherwise the B operand (rows = pixels, columns = output channels).
template <int PIXB, bool SWAP>
__device__ __forceinline__ void conv_two_rows(f32x16 (&acc)[2], const bf16x8 (&wh)[18], const bf16x8 (&wl)[18],
                                              const unsigned char* ph, const unsigned char* pl, int pix0, int chunk0) {
    constexpr int NF = 24, DEPTH = 3;
    bf16x8 fh[DEPTH], fl[DEPTH];
    // chunk0 has bit 1 clear, so the second 16-channel block of a pixel (chunk0 + 2) sits at the first one's offset ^ 32
    auto fetch = [&](int f, int slot) {
        const int pr = f / 6, dx = (f >> 1) % 3, kb = f & 1;
        const int off = pix_off<PIXB>(pix0 + pr * PW + dx, chunk0) ^ (kb << 5);
        fh[slot] = *reinterpret_cast<const bf16x8*>(ph + off);
        fl[slot] = *reinterpret_cast<const bf16x8*>(pl + off);
    };
    fetch(0, 0);
    fetch(1, 1);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + 2 < NF) fetch(f + 2, (f + 2) % DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        const int s = f % DEPTH, pr = f / 6, dx = (f >> 1) % 3, kb = f & 1;
        if (pr <= 2) {
            const int st = (pr * 3 + dx) * 2 + kb;
            if constexpr (SWAP) {
                acc[0] = MFMA16(wl[st], fh[s], acc[0]);
                acc[0] = MFMA16(wh[st], fl[s], acc[0]);
                acc[0] = MFMA16(wh[st], fh[s], acc[0]);
            } else {
                acc[0] = MFMA16(fl[s], wh[st], acc[0]);
                acc[0] = MFMA16(fh[s], wl[st], acc[0]);
                acc[0] = MFMA16(fh[s], wh[st], acc[0]);
            }
        }
        if (pr >= 1) {
            const int st = ((pr - 1) * 3 + dx) * 2 + kb;
            if constexpr (SWAP) {
                acc[1] = MFMA16(wl[st], fh[s], acc[1]);
                acc[1] = MFMA16(wh[st], fl[s], acc[1]);
                acc[1] = MFMA16(wh[st], fh[s], acc[1]);
            } else {
                acc[1] = MFMA16(fl[s], wh[st], acc[1]);
                acc[1] = MFMA16(fh[s], wl[st], acc[1]);
                acc[1] = MFMA16(fh[s], wh[st], acc[1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 + conv2 + pool
// ------------------------------------------------------------------------------------------------------------------
constexpr int P12_ROWS = 10;                       // conv1 rows 0 .. 9 feed conv2 rows 0 .. 7 (the ninth conv2 row is dropped by the pool)
constexpr int P12_PIX = P12_ROWS * PW;             // 340 patch pixels
constexpr int P12_PIECE = P12_PIX * 64;            // bytes per piece
constexpr int P12_BUF = 2 * P12_PIECE;             // hi + lo
constexpr int P12_TILES = (P12_PIX + 31) / 32;     // 11 conv1 tiles of 32 pixels
constexpr int P12_TPW = (P12_TILES + 3) / 4;       // per producer wave
constexpr int XR = P12_ROWS + 2, XW = PW + 2;      // raw input region of a task: rows -1 .. 10, columns x0 - 1 .. x0 + 34
constexpr int XIN = XR * XW;                       // 432 floats
constexpr int XIN_DMA = (XIN + 63) / 64;           // 256-byte LDS-DMA pieces
constexpr int XIN_BYTES = ((XIN * 4 + 255) / 256) * 256;
constexpr size_t LDS12 = 2 * (size_t)XIN_BYTES + 2 * (size_t)P12_BUF;   // two raw-input buffers (first: LDS-DMA targets), two patch buffers

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
// global -> LDS without registers (global_load_lds_dword / _dwordx4): the LDS address is `l` + lane x SIZE, `l` wave-uniform
__device__ __forceinline__ void dma4_to_lds(const void* g, void* l) { __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 4, 0, 0); }
__device__ __forceinline__ void dma16_to_lds(const void* g, void* l) { __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }

// Per iteration k of a workgroup (its tasks t_0, t_1, ...; ONE barrier per iteration):
//   producers  raw input of t_(k+2): HBM -> LDS by LDS-DMA (zeros outside the window come from a zero buffer);
//              conv1 of t_(k+1) from the raw input that landed an iteration ago -> patch buffer (k + 1) & 1
//   consumers  conv2 + pool of t_k from patch buffer k & 1
__global__ __launch_bounds__(512) void k_conv12_ws(const float* __restrict__ x, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, const bf16x8* __restrict__ w2s,
                                                   const float* __restrict__ b2, __bf16* __restrict__ out,
                                                   const float* __restrict__ zeros, int Hin, int Win, int xtiles, int64_t ntask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const ldsP = lds + 2 * XIN_BYTES;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: roles, LDS-DMA targets)
    const int i = lane & 31, h = lane >> 5;
    const int Wout = (Win - 2) / 2;
    const int64_t G = gridDim.x;
    const int64_t nloc = (ntask - blockIdx.x + G - 1) / G;     // tasks of this workgroup: blockIdx.x + k G

    if (wave < 4) {
        // ---------------- consumers: conv2 rows 2 wave, 2 wave + 1 ----------------
        bf16x8 wh[18], wl[18];
        {
            const bf16x8* ph = w2s + h * C2 + i;               // [piece][tap][kb][h][cout]
            const bf16x8* pl = ph + 18 * 2 * C2;
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                wh[st] = ph[st * 2 * C2];
                wl[st] = pl[st * 2 * C2];
            }
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                pin(wh[st]);
                pin(wl[st]);
            }
        }
        const float bias = b2[i];
        for (int64_t k = -2; k < nloc; ++k) {
            if (k >= 0) {
                const int64_t t = blockIdx.x + k * G;
                const int64_t win = t / xtiles;
                const int x0 = 32 * (int)(t - win * xtiles);
                const unsigned char* ph = ldsP + (k & 1) * P12_BUF;
                f32x16 acc[2];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[0][q] = 0.f;
                    acc[1][q] = 0.f;
                }
                conv_two_rows<64, false>(acc, wh, wl, ph, ph + P12_PIECE, (2 * wave) * PW + i, h);
                // 2 x 2 pool inside the lane: register pair (q, q + 1) = columns (2 t, 2 t + 1), the two accumulators = the two rows
                __bf16* o = out + ((win * 4 + wave) * (int64_t)Wout) * 64 + i;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int q = 2 * kk;
                    const int px = (x0 + (q & 3) + 8 * (q >> 2) + 4 * h) >> 1;
                    const float m = fmaxf(fmaxf(acc[0][q], acc[0][q + 1]), fmaxf(acc[1][q], acc[1][q + 1]));
                    const float v = fmaxf(m + bias, 0.f);
                    const __bf16 vh = (__bf16)v;
                    const __bf16 vl = (__bf16)(v - (float)vh);
                    if (px < Wout) {
                        o[(int64_t)px * 64] = vh;
                        o[(int64_t)px * 64 + 32] = vl;
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ---------------- producers ----------------
        const int pw = wave - 4;
        // W1^T as the A operand: row = channel i, k = tap (0..8), 9 = bias (times a constant 1), 10..15 = 0
        bf16x8 w1h, w1l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 8 * h + j;
            const float v = tap < 9 ? w1[(tap < 9 ? tap : 0) * C1 + i] : tap == 9 ? b1[i] : 0.f;
            const __bf16 vh = (__bf16)v;
            w1h[j] = vh;
            w1l[j] = (__bf16)(v - (float)vh);
        }
        pin(w1h);
        pin(w1l);
        for (int64_t k = -2; k < nloc; ++k) {
            if (k + 2 < nloc) {
                // raw input region of task t_(k+2): element e = (row r, column c) of the 12 x 36 region, one dword per lane
                const int64_t t = blockIdx.x + (k + 2) * G;
                const int64_t win = t / xtiles;
                const int x0 = 32 * (int)(t - win * xtiles);
                const float* img = x + win * (int64_t)Hin * Win;
                unsigned char* xb = lds + (k & 1) * XIN_BYTES;
#pragma unroll
                for (int m = 0; m < (XIN_DMA + 3) / 4; ++m) {
                    const int piece = pw + 4 * m;              // wave-uniform
                    if (piece < XIN_DMA) {
                        const int e = 64 * piece + lane;
                        const int r = e / XW, c = e - r * XW;
                        const int yi = r - 1, xi = x0 - 1 + c;
                        const float* src = (yi >= 0 && yi < Hin && xi >= 0 && xi < Win) ? img + yi * Win + xi : zeros;
                        if (e < XIN) dma4_to_lds(src, xb + 256 * piece);
                    }
                }
            }
            if (k + 1 >= 0 && k + 1 < nloc) {
                // conv1 of task t_(k+1): lane (i, h) supplies taps 8 h + j of pixel 32 tile + i
                const float* xin = reinterpret_cast<const float*>(lds + ((k + 1) & 1) * XIN_BYTES);
                unsigned char* ph = ldsP + ((k + 1) & 1) * P12_BUF;
                unsigned char* pl = ph + P12_PIECE;
#pragma unroll
                for (int tt = 0; tt < P12_TPW; ++tt) {
                    const int tile = pw + 4 * tt;
                    if (tile >= P12_TILES) continue;           // wave-uniform
                    const int e = 32 * tile + i;
                    const int ec = min(e, P12_PIX - 1);
                    const int pr = ec / PW, pc = ec - pr * PW;
                    const float* xp = xin + pr * XW + pc;      // taps (dy, dx) at xp[dy * XW + dx]
                    float xv[8];
                    xv[0] = xp[h ? 2 * XW + 2 : 0];            // tap 0, or tap 8 in the upper half
#pragma unroll
                    for (int j = 1; j < 8; ++j) {
                        const float v = xp[(j / 3) * XW + (j % 3)];
                        xv[j] = h ? (j == 1 ? 1.f : 0.f) : v;  // upper half: the bias slot, then zeros
                    }
                    bf16x8 xh, xl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const __bf16 vh = (__bf16)xv[j];
                        xh[j] = vh;
                        xl[j] = (__bf16)(xv[j] - (float)vh);
                    }
                    f32x16 a;
#pragma unroll
                    for (int q = 0; q < 16; ++q) a[q] = 0.f;
                    a = MFMA16(w1l, xh, a);
                    a = MFMA16(w1h, xl, a);
                    a = MFMA16(w1h, xh, a);
                    if (e < P12_PIX) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {          // channels 8 g + 4 h .. + 3 of pixel e
                            bf16x4 vh, vl;
                            split4(relu(a[4 * g]), relu(a[4 * g + 1]), relu(a[4 * g + 2]), relu(a[4 * g + 3]), vh, vl);
                            const int off = pix_off<64>(e, g) + 8 * h;
                            *reinterpret_cast<bf16x4*>(ph + off) = vh;
                            *reinterpret_cast<bf16x4*>(pl + off) = vl;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3 + conv4 + pool
// ------------------------------------------------------------------------------------------------------------------
constexpr int PA_PIX = 6 * PW;                     // conv3 input patch: pooled conv2 rows -1 .. 4, columns c0 - 1 .. c0 + 32
constexpr int PA_PIECE = PA_PIX * 64;
constexpr int PA_BUF = 2 * PA_PIECE;
constexpr int PB_PIX = 4 * PW;                     // conv4 input patch: conv3 rows 0 .. 3, columns c0 .. c0 + 33
constexpr int PB_PIECE = PB_PIX * 128;
constexpr int PB_BUF = 2 * PB_PIECE;
constexpr int X_BUF = 2 * 8 * 64 * 16;             // partial sums of the second K half: 2 output tiles x 8 register quads x 64 lanes x 16 B
constexpr int PA_CHUNKS = 2 * PA_PIX * 4;          // 16-byte chunks of a patch A buffer
constexpr int PA_DMA = (PA_CHUNKS + 63) / 64;      // 1 KB LDS-DMA pieces of a patch A buffer
constexpr int BIAS3_BYTES = 2 * 2 * 16 * 4;        // conv3 biases in accumulator order: [output tile][lane half][register]
constexpr size_t LDS34 = 2 * (size_t)PA_BUF + 2 * (size_t)PB_BUF + 2 * (size_t)X_BUF + BIAS3_BYTES;
static_assert(LDS34 <= 160 * 1024, "one workgroup per CU");

__global__ __launch_bounds__(512) void k_conv34_ws(const uint4* __restrict__ in, const bf16x8* __restrict__ w3s,
                                                   const float* __restrict__ b3, const bf16x8* __restrict__ w4s,
                                                   const float* __restrict__ b4, float* __restrict__ out,
                                                   const uint4* __restrict__ zeros, int Win, int xtiles, int64_t ntile) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 2 * PA_BUF;
    unsigned char* const ldsX = ldsB + 2 * PB_BUF;
    float* const ldsBias = reinterpret_cast<float*>(ldsX + 2 * X_BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int Wp = (Win - 2) / 2;                             // pooled conv4 width
    const int64_t G = gridDim.x;
    // tiles of this workgroup: blockIdx.x + j G, j = 0 .. nloc - 1
    const int64_t nloc = (ntile - blockIdx.x + G - 1) / G;

    // the columns of patch B that only conv4's discarded outputs read are never written: clear both buffers once
    for (int e = tid; e < (int)(2 * PB_BUF / 16); e += 512) reinterpret_cast<uint4*>(ldsB)[e] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 64) ldsBias[tid] = b3[(tid >> 5) * 32 + (tid & 3) + 8 * ((tid & 15) >> 2) + 4 * ((tid >> 4) & 1)];
    __syncthreads();

    if (wave < 4) {
        // ---------------- conv3: output tile nt (32 of the 64 channels), rows 2 rp, 2 rp + 1 ----------------
        const int nt = wave & 1, rp = wave >> 1;
        bf16x8 wh[18], wl[18];
        {
            const bf16x8* ph = w3s + h * C3 + nt * 32 + i;    // [piece][tap][kb][h][cout]
            const bf16x8* pl = ph + 18 * 2 * C3;
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                wh[st] = ph[st * 2 * C3];
                wl[st] = pl[st * 2 * C3];
            }
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                pin(wh[st]);
                pin(wl[st]);
            }
        }
        const float4* bias4 = reinterpret_cast<const float4*>(ldsBias + (nt * 2 + h) * 16);   // (16 registers too many to hold)
        for (int64_t j = -2; j <= nloc; ++j) {
            const int64_t jt = j + 1;                         // conv3 works one tile ahead of conv4
            if (jt >= 0 && jt < nloc) {
                const unsigned char* pa = ldsA + (jt & 1) * PA_BUF;
                unsigned char* pbh = ldsB + (jt & 1) * PB_BUF;
                unsigned char* pbl = pbh + PB_PIECE;
                f32x16 acc[2];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[0][q] = 0.f;
                    acc[1][q] = 0.f;
                }
                conv_two_rows<64, true>(acc, wh, wl, pa, pa + PA_PIECE, (2 * rp) * PW + i, h);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int pb = (2 * rp + mt) * PW + i;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {             // channels 32 nt + 8 g + 4 h .. + 3 of pixel pb
                        bf16x4 vh, vl;
                        const float4 bq = bias4[g];
                        split4(relu(acc[mt][4 * g] + bq.x), relu(acc[mt][4 * g + 1] + bq.y), relu(acc[mt][4 * g + 2] + bq.z),
                               relu(acc[mt][4 * g + 3] + bq.w), vh, vl);
                        const int off = pix_off<128>(pb, nt * 4 + g) + 8 * h;
                        *reinterpret_cast<bf16x4*>(pbh + off) = vh;
                        *reinterpret_cast<bf16x4*>(pbl + off) = vl;
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ---------------- loader + conv4: output tile nt, K half kh (input channels 32 kh .. 32 kh + 31) ----------------
        const int cw = wave - 4, nt = cw & 1, kh = cw >> 1;
        bf16x8 wh[18], wl[18];
        {
            const bf16x8* ph = w4s + h * C4 + nt * 32 + i;    // [piece][tap][kb (4)][h][cout]
            const bf16x8* pl = ph + 36 * 2 * C4;
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int it = (st >> 1) * 4 + 2 * kh + (st & 1);
                wh[st] = ph[it * 2 * C4];
                wl[st] = pl[it * 2 * C4];
            }
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                pin(wh[st]);
                pin(wl[st]);
            }
        }
        const float bias = b4[nt * 32 + i];
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[0][q] = 0.f;
            acc[1][q] = 0.f;
        }
        for (int64_t j = -2; j <= nloc; ++j) {
            // (1) first K half: the partial sums the second half left in LDS an iteration ago complete tile j - 1
            if (kh == 0 && j - 1 >= 0 && j - 1 < nloc) {
                const int64_t tile = blockIdx.x + (j - 1) * G;
                const int64_t win = tile / xtiles;
                const int c0 = T34 * (int)(tile - win * xtiles);
                const uint4* xs = reinterpret_cast<const uint4*>(ldsX + ((j - 1) & 1) * X_BUF) + nt * (8 * 64) + lane;
#pragma unroll
                for (int r4 = 0; r4 < 8; ++r4) {
                    const uint4 v = xs[r4 * 64];
                    const int mt = r4 >> 2, q = 4 * (r4 & 3);
                    acc[mt][q] += __uint_as_float(v.x);
                    acc[mt][q + 1] += __uint_as_float(v.y);
                    acc[mt][q + 2] += __uint_as_float(v.z);
                    acc[mt][q + 3] += __uint_as_float(v.w);
                }
                float* o = out + win * (int64_t)Wp * C4 + nt * 32 + i;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = 2 * k;
                    const int xl = (q & 3) + 8 * (q >> 2) + 4 * h;    // even conv4 column inside the tile
                    const int pxp = (c0 + xl) >> 1;
                    const float m = fmaxf(fmaxf(acc[0][q], acc[0][q + 1]), fmaxf(acc[1][q], acc[1][q + 1]));
                    if (xl < T34 && pxp < Wp) o[(int64_t)pxp * C4] = fmaxf(m + bias, 0.f);
                }
            }
            // (2) tile j + 2 from HBM into the patch A buffer conv3 read in the previous iteration, by LDS-DMA: 16-byte slot s of
            // the buffer = (piece, pixel, position) <- chunk position ^ swizzle(pixel) of that pixel's piece (the XOR is its own
            // inverse: the readers apply the same one); pixels outside the image come from the zero buffer
            if (j + 2 < nloc) {
                const int64_t tile = blockIdx.x + (j + 2) * G;
                const int64_t win = tile / xtiles;
                const int c0 = T34 * (int)(tile - win * xtiles);
                unsigned char* pa = ldsA + (j & 1) * PA_BUF;
                const uint4* img = in + win * 4 * (int64_t)Win * 8;
                int ln = lane;
                asm volatile("" : "+v"(ln));                  // (slot coordinates formed per tile, not held across the matrix loop)
#pragma unroll
                for (int m = 0; m < (PA_DMA + 3) / 4; ++m) {
                    const int dp = cw + 4 * m;                // wave-uniform
                    if (dp < PA_DMA) {
                        const int sl = 64 * dp + ln;
                        const int piece = sl >= PA_PIX * 4 ? 1 : 0, rem = sl - piece * (PA_PIX * 4);
                        const int pixel = rem >> 2, c = (rem & 3) ^ ((pixel >> 2) & 3);
                        const int r = pixel / PW, pc = pixel - r * PW;
                        const int yi = r - 1, xi = c0 - 1 + pc;
                        const uint4* src = (yi >= 0 && yi < 4 && xi >= 0 && xi < Win) ? img + (yi * Win + xi) * 8 + piece * 4 + c : zeros;
                        if (sl < PA_CHUNKS) dma16_to_lds(src, pa + 1024 * dp);
                    }
                }
            }
            // (3) conv4 of tile j, this wave's K half
            if (j >= 0 && j < nloc) {
                const unsigned char* pbh = ldsB + (j & 1) * PB_BUF;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[0][q] = 0.f;
                    acc[1][q] = 0.f;
                }
                conv_two_rows<128, false>(acc, wh, wl, pbh, pbh + PB_PIECE, i, 4 * kh + h);
                if (kh == 1) {
                    uint4* xs = reinterpret_cast<uint4*>(ldsX + (j & 1) * X_BUF) + nt * (8 * 64) + lane;
#pragma unroll
                    for (int r4 = 0; r4 < 8; ++r4) {
                        const int mt = r4 >> 2, q = 4 * (r4 & 3);
                        xs[r4 * 64] = make_uint4(__float_as_uint(acc[mt][q]), __float_as_uint(acc[mt][q + 1]),
                                                 __float_as_uint(acc[mt][q + 2]), __float_as_uint(acc[mt][q + 3]));
                    }
                }
            }
            __syncthreads();
        }
    }
}

}  // namespace

// conv1 .. conv4 + pools of n windows: x (n, H1, W1) float32 -> a2s (n, 4, W1/2 - 1, [hi 32 | lo 32]) bf16 (scratch) ->
// a4 (n, 1, Wp2, 64) float32. Only for windows with four pooled rows after conv2 (f2_cnn_ws_supported).
bool f2_cnn_ws_supported(int rows, int channels) {
    const int Hp1 = (rows - 2) / 2, Wp1 = (channels - 2) / 2;
    return Hp1 == 4 && Wp1 >= 5;
}

int f2_launch_cnn_ws(f2_ctx* ctx, const f2_cnn* cnn, const float* d_x, int64_t n, void* a2s, float* a4) {
    const int H1 = cnn->rows, W1 = cnn->channels;
    const int Wo = W1 - 2, Wp1 = Wo / 2, Wp2 = (Wp1 - 2) / 2;
    const int grid_max = ctx->num_cus > 0 ? ctx->num_cus : 256;
    {
        const int xtiles = ((Wo / 2) * 2 + 31) / 32;
        const int64_t ntask = n * xtiles;
        static_assert(LDS12 <= 160 * 1024, "one workgroup per CU");
        F2_HIP(ctx, hipFuncSetAttribute((const void*)k_conv12_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS12));
        const unsigned grid = (unsigned)(ntask < grid_max ? ntask : grid_max);
        hipLaunchKernelGGL(k_conv12_ws, dim3(grid), dim3(512), LDS12, ctx->stream, d_x, cnn->t(0), cnn->t(1),
                           (const bf16x8*)(cnn->blob16 + cnn->off16[0]), cnn->t(3), (__bf16*)a2s, (const float*)cnn->zeros, H1, W1,
                           xtiles, ntask);
        F2_HIP(ctx, hipGetLastError());
    }
    {
        const int xtiles = (2 * Wp2 + T34 - 1) / T34;
        const int64_t ntile = n * xtiles;
        F2_HIP(ctx, hipFuncSetAttribute((const void*)k_conv34_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS34));
        const unsigned grid = (unsigned)(ntile < grid_max ? ntile : grid_max);
        hipLaunchKernelGGL(k_conv34_ws, dim3(grid), dim3(512), LDS34, ctx->stream, (const uint4*)a2s,
                           (const bf16x8*)(cnn->blob16 + cnn->off16[1]), cnn->t(5), (const bf16x8*)(cnn->blob16 + cnn->off16[2]),
                           cnn->t(7), a4, (const uint4*)cnn->zeros, Wp1, xtiles, ntile);
        F2_HIP(ctx, hipGetLastError());
    }
    return F2_OK;
}
