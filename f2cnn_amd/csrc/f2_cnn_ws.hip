// K4, weight-stationary form of the split-fp16 convolutions (round 4; fp16 pieces and per-layer scales since round 5:
// f2_cnn_split.h - where this file says "split" it means two fp16 pieces) for the reference's window shape (10 or 11 rows:
// four pooled rows after conv2). Reference: architecture scripts/CNN/Training.py:93-114, predict scripts/CNN/Evaluating.py:84-87.
//
// Why. k_conv12_h16x3 / k_conv34_h16x3 (f2_cnn.hip) issue MFMAs 37 % of the time: every 32 x 32 x 16 product loads both of
// its operands again - the activations from LDS and the weights through the vector L1, 2 KB per wave and step, 85 B/clk/CU
// against the 64 B/clk the L1 delivers - and conv1 runs as 36 float32 FMAs per pixel and channel quad in front of the matrix loop,
// in phases that all waves of a workgroup go through together.
//
// Here every wave keeps the weights of its role in registers for the whole launch (9 taps x 32 input channels x 32 outputs x 2
// fp16 pieces = 144 VGPRs; 256 registers per wave, one 8-wave workgroup per CU, persistent over tiles), with ONE barrier per
// tile. What the other instructions cost was measured, not assumed (tools/ws_stamps.py: in-kernel stamps; tools/ubench/
// mfma_valu_coissue.hip, mfma_fillers.hip; SQ counters under profiles/): beside a partner wave's matrix loop a wave issues only
// ~2 VALU instructions per MFMA whatever the wave priorities, while up to ~8 per MFMA ride in the gaps of a wave's OWN matrix
// loop for +16 % loop time - so the kernels are counted in instructions: no integer divisions, no per-store branches, per-tile
// address arithmetic held in registers, ReLU as one v_med3, LDS-DMA for every global -> LDS copy.
//
//   k_conv12_ws   tile = (window, 32 conv2 columns), all 8 conv2 rows the pool keeps. Wave w: conv2 of rows 2 (w & 3), + 1 x
//                 columns 16 (w >> 2) .. + 15 of tile t (the 32 x 32 accumulator's row index = (row, column): the 2 x 2 max-pool
//                 stays inside the lane). Riding behind the steps of that matrix loop: conv1 of tile t + 1 for one or two
//                 blocks of 32 patch pixels ON THE MATRIX CORES - out^T (32 channels x 32 pixels) = W1^T (32 x 16: nine taps,
//                 the bias against a constant 1, zeros) x taps^T (16 x 32 pixels), three split-fp16 MFMAs; with the channels on
//                 the accumulator's row index a lane holds four consecutive channels of its pixel, so ReLU + split + two 8-byte
//                 LDS stores per quad write the conv2 input patch - no longer the oracle's fmaf chain bit for bit, held to the
//                 same 2e-5 / referee rule; the LDS-DMA of the raw input of tile t + 2 (zeros outside the window from a zero
//                 buffer); bias + ReLU + split + stores of the pooled outputs of tile t - 1. Output already split in two fp16
//                 pieces ([window][4][16 x tiles][hi 32 | lo 32]): conv3's staging is a copy.
//   k_conv34_ws   tile = (window, 30 conv4 columns). waves 0-3: LDS-DMA of tile t + 2 from HBM into the free patch A, conv3 (output
//                 tile nt, row pair: a patch row fetched from LDS serves tap row dy of one output row and dy - 1 of the other,
//                 48 ds_read_b128 per 108 MFMAs) of tile t + 1 from patch A into patch B (64 channels per pixel, both pieces;
//                 accumulator rows = output channels as above); waves 4-7: the two K halves of tile t - 1 meet (first half: sum,
//                 pool, bias, store), then conv4 (output tile nt, K half) of tile t from patch B.
//
//   k_dense1_ws   2 x 96 windows x 6 output tiles per 12-wave workgroup (one per CU, three waves on every SIMD), weights streamed
//                 two 16-byte loads per lane and K step into a ring, activations split on their way into LDS and fetched a K step
//                 ahead of their MFMAs; every load of the K loop waited for by hand.
//
// Layout of an MFMA operand (tools/ubench/mfma_bf16_layout.hip, mfma_f16_split.hip: the two forms share it): lane (i, h) supplies k = 8 h .. 8 h + 7 of row / column i;
// accumulator register q of lane (i, h) is row (q & 3) + 8 (q >> 2) + 4 h, column i.
#include <type_traits>

#include "f2_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C1 = 32, C2 = 32, C3 = 64, C4 = 64;
constexpr int PW = 34;            // patch width: 32 output columns + 2
constexpr int T34 = 30;           // conv4 output columns per tile

// LDS images of the activation patches. A matrix step takes 16 input channels (kb = which 16) of 32 pixels: lane (i, h) reads the
// 16 bytes of channels 16 kb + 8 h .. + 7 of its pixel. Every piece (hi / lo) of a patch is stored as PLANES of 16-channel
// groups so that kb is a compile-time byte offset (one address register per tap instead of two), and inside a plane the
// 16-byte chunks are XOR-swizzled so that ds_read_b128 (four groups of 16 lanes, sixteen 16-byte slots per LDS cycle) is
// conflict-free for the access pattern of its reader (brute force: tools/lds_swizzle_check.py):
//   32-channel patches: two planes of 32 bytes per pixel, half h at p * 32 + ((h ^ s) << 4) with
//     s = (p >> 3) & 1   conv3's input, read 32 consecutive pixels of a row at a time                     (a32_off)
//     s = row & 1        conv2's input, read two rows x 16 columns at a time (the pool stays in the lane)  (c32_off)
//   64-channel patch (conv4's input): two planes of 64 bytes per pixel (channels 32 kh + 16 kb + 8 h: chunk q = 2 kh + h of plane
//     kb) at p * 64 + ((q ^ ((p >> 2) & 3)) << 4)                                                          (b64_off)
__device__ __forceinline__ int a32_off(int p, int h) { return p * 32 + ((h ^ ((p >> 3) & 1)) << 4); }
__device__ __forceinline__ int c32_off(int row, int col, int h) { return (row * PW + col) * 32 + ((h ^ (row & 1)) << 4); }
__device__ __forceinline__ int b64_off(int p, int q) { return p * 64 + ((q ^ ((p >> 2) & 3)) << 4); }

// ReLU as ONE instruction: integer max with 0 on the bit pattern (negative floats are negative integers; fmaxf and v_med3 alike come
// out of this compiler as a canonicalising v_max(v, v) followed by v_max(0, v))
__device__ __forceinline__ float relu(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

// a value the compiler must keep in a register from here on (it cannot re-load or re-derive it inside the tile loop)
__device__ __forceinline__ void pin(h16x8& v) { asm volatile("" : "+v"(v)); }

// Issue priority of this wave on its SIMD: high in the VALU / LDS phases, low in the matrix loop (which needs one issue slot per
// 32 cycles). Measured effect: none either way (tools/ws_stamps.py); kept because it is the documented intent.
#define PRIO_VALU() __builtin_amdgcn_s_setprio(3)
#define PRIO_MATRIX() __builtin_amdgcn_s_setprio(0)

// tile index -> (window, tile in window) without a division: q = floor(t / d) = mulhi(t, ceil(2^32 / d)) for t < 2^32 / d
struct tile_div {
    unsigned d, mul;
};
__device__ __forceinline__ void tile_split(unsigned t, tile_div td, unsigned& q, unsigned& r) {
    q = td.d == 1 ? t : __umulhi(t, td.mul);
    r = t - q * td.d;
}

// Diagnostic build (-DF2_WS_STAMPS, tools/build_variant.sh): s_memtime at phase boundaries of iterations 8 .. 39 of the first
// eight workgroups, [workgroup][wave][iteration][slot]; slot 7 = s_memrealtime at the start of the iteration (clock estimate).
#ifdef F2_WS_STAMPS
#define WS_STAMP_ARG , unsigned long long* stamps
#define WS_STAMP(it, slot)                                                                                              \
    do {                                                                                                                \
        if (stamps && blockIdx.x < 8 && wave < 8 && (it) >= 8 && (it) < 40 && lane == 0) {                                          \
            unsigned long long* sp_ = stamps + ((((size_t)blockIdx.x * 8 + wave) * 32 + ((it) - 8)) * 8);               \
            sp_[slot] = __builtin_amdgcn_s_memtime();                                                                   \
            if ((slot) == 0) sp_[7] = __builtin_amdgcn_s_memrealtime();                                                 \
        }                                                                                                               \
    } while (0)
#else
#define WS_STAMP_ARG
#define WS_STAMP(it, slot)
#endif

// Two output rows x 32 columns x 32 output channels of a 3 x 3 convolution over 32 input channels (18 steps of 16): the
// patch rows r0 .. r0 + 3 are fetched once (hi and lo piece of (pixel, 16 channels): two ds_read_b128), row pr serving tap row
// dy = pr of output row 0 and dy = pr - 1 of output row 1. off(pr, dx) = byte offset of this lane's 16 bytes in plane 0 for patch
// row pr, tap column dx; PLANE = bytes between the two 16-channel planes. SWAP: the weights are the A operand (accumulator rows =
// output channels, columns = pixels), otherwise the B operand (rows = pixels, columns = output channels).
// fill(f) rides behind the MFMAs of fragment f (see conv_one_tile); output row 0 is complete once fragment 17 is done.
template <bool SWAP, int DEPTH, int PLANE, class OFF, class FILL>
__device__ __forceinline__ void conv_two_rows(f32x16 (&acc)[2], const h16x8 (&wh)[18], const h16x8 (&wl)[18],
                                              const unsigned char* ph, const unsigned char* pl, OFF off, FILL fill) {
    constexpr int NF = 24;
    h16x8 fh[DEPTH], fl[DEPTH];
    auto fetch = [&](int f, int slot) {
        const int pr = f / 6, dx = (f >> 1) % 3, kb = f & 1;
        const int o = off(pr, dx) + kb * PLANE;
        fh[slot] = *reinterpret_cast<const h16x8*>(ph + o);
        fl[slot] = *reinterpret_cast<const h16x8*>(pl + o);
    };
#pragma unroll
    for (int f = 0; f < DEPTH - 1; ++f) fetch(f, f);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + DEPTH - 1 < NF) fetch(f + DEPTH - 1, (f + DEPTH - 1) % DEPTH);
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
        fill(f);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One 32 x 32 accumulator of the same convolution: off(dy, dx) = byte offset in plane 0 of this lane's pixel for tap (dy, dx);
// 18 steps, two ds_read_b128 + three MFMAs each, DEPTH - 1 steps requested ahead.
// fill(f) is placed behind the MFMAs of step f: other work of the SAME wave rides in the gaps of its own matrix loop (up to ~8
// VALU instructions per MFMA for +16 % loop time, tools/ubench/mfma_fillers.hip; a partner wave's VALU beside the loop gets ~2 per
// MFMA, tools/ubench/mfma_valu_coissue.hip).
template <int DEPTH, int PLANE, class OFF, class FILL>
__device__ __forceinline__ void conv_one_tile(f32x16& acc, const h16x8 (&wh)[18], const h16x8 (&wl)[18],
                                              const unsigned char* ph, const unsigned char* pl, OFF off, FILL fill) {
    constexpr int NF = 18;
    h16x8 fh[DEPTH], fl[DEPTH];
    auto fetch = [&](int f, int slot) {
        const int tap = f >> 1, kb = f & 1, dy = tap / 3, dx = tap - 3 * dy;
        const int o = off(dy, dx) + kb * PLANE;
        fh[slot] = *reinterpret_cast<const h16x8*>(ph + o);
        fl[slot] = *reinterpret_cast<const h16x8*>(pl + o);
    };
#pragma unroll
    for (int f = 0; f < DEPTH - 1; ++f) fetch(f, f);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f + DEPTH - 1 < NF) fetch(f + DEPTH - 1, (f + DEPTH - 1) % DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        const int s = f % DEPTH;
        acc = MFMA16(fl[s], wh[f], acc);
        acc = MFMA16(fh[s], wl[f], acc);
        acc = MFMA16(fh[s], wh[f], acc);
        fill(f);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 + conv2 + pool
// ------------------------------------------------------------------------------------------------------------------
constexpr int P12_ROWS = 10;                       // conv1 rows 0 .. 9 feed conv2 rows 0 .. 7 (the ninth conv2 row is dropped by the pool)
constexpr int P12_PIX = P12_ROWS * PW;             // 340 patch pixels
constexpr int P12_PLANE = P12_PIX * 32;            // one 16-channel plane of a piece
constexpr int P12_PIECE = 2 * P12_PLANE;           // bytes per piece
constexpr int P12_BUF = 2 * P12_PIECE;             // hi + lo
constexpr int P12_TILES = (P12_PIX + 31) / 32;     // 11 conv1 blocks of 32 pixels: wave w takes block w, waves 0-2 also block 8 + w
constexpr int XR = P12_ROWS + 2, XW = PW + 2;      // raw input region of a task: rows -1 .. 10, columns x0 - 1 .. x0 + 34
constexpr int XIN = XR * XW;                       // 432 floats
static_assert((XIN + 63) / 64 == 7, "LDS-DMA pieces of the raw input: waves 4-7 take piece w - 4, waves 4-6 also piece w");
constexpr int XIN_BYTES = ((XIN * 4 + 255) / 256) * 256;
constexpr int CTAB = 2 * XW + 2;                   // floats: 1, 0, 0, ... - what the upper lane half reads in place of taps 1 .. 7,
constexpr int CTAB_BYTES = ((CTAB * 4 + 255) / 256) * 256;   // one copy behind EACH raw-input buffer (same offset from either base)
constexpr int XIN_STRIDE = XIN_BYTES + CTAB_BYTES;
constexpr int W1_BYTES = 2 * 64 * 16;               // conv1's A operand (W1^T, bias row), hi and lo piece, 16 bytes per lane
constexpr size_t LDS12 = 2 * (size_t)XIN_STRIDE + 2 * (size_t)P12_BUF + W1_BYTES;   // (raw input + table) x 2 (LDS-DMA targets first), patches x 2, W1

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// global -> LDS without registers (global_load_lds_dword / _dwordx4; LDS address = `lds_addr` + lane x size, wave-uniform), as
// inline asm: with the builtin in the kernel, hipcc (ROCm 7.2) stops counting LDS reads and waits `lgkmcnt(0)` in front of the
// MFMAs - the fragment just requested included - which took the matrix loops from 34 to 52 cycles per MFMA. Hidden from the
// compiler, the loads are waited for by hand: dma_wait() before the barrier that hands the buffer over.
__device__ __forceinline__ unsigned lds_address(const void* p) { return (unsigned)(uintptr_t)(lds_ptr_t)p; }
__device__ __forceinline__ void dma4_to_lds(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16_to_lds(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Per iteration k of a workgroup (its tasks t_0, t_1, ...; ONE barrier per iteration), every wave: the matrix loop of conv2 of
// its 2 x 16 pixels of t_k from patch buffer k & 1 and, behind its steps: its LDS-DMA pieces of the raw input of t_(k+2), conv1
// of its one or two pixel blocks of t_(k+1) (from the raw input that landed an iteration ago, into patch buffer (k + 1) & 1), the
// stores of its pooled outputs of t_(k-1). The first two and the last two iterations run the same pieces one after the other.
__global__ __launch_bounds__(512) void k_conv12_ws(const float* __restrict__ x, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, const h16x8* __restrict__ w2s,
                                                   const float* __restrict__ b2 /* x sa_3 */, _Float16* __restrict__ out,
                                                   const float* __restrict__ zeros, int Hin, int Win, int Wa, tile_div xt,
                                                   unsigned ntask, f2_split_scales S WS_STAMP_ARG) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const ldsP = lds + 2 * XIN_STRIDE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: roles, LDS-DMA targets)
    const int i = lane & 31, h = lane >> 5;
    const unsigned G = gridDim.x;
    const int nloc = (int)((ntask - blockIdx.x + G - 1) / G);  // tasks of this workgroup: blockIdx.x + k G
    const int rp = wave & 3, ch = wave >> 2;

    if (threadIdx.x < 2 * CTAB) {
        const int b = threadIdx.x >= CTAB ? 1 : 0, e = threadIdx.x - b * CTAB;
        reinterpret_cast<float*>(lds + b * XIN_STRIDE + XIN_BYTES)[e] = e == 0 ? 1.f : 0.f;
    }

    // conv2 weights of every wave: B operand, [piece][tap][kb][h][cout]
    h16x8 wh[18], wl[18];
    {
        const h16x8* ph = w2s + h * C2 + i;
        const h16x8* pl = ph + 18 * 2 * C2;
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
    float bias = b2[i];
    asm volatile("" : "+v"(bias));         // (settled here: a load still "pending" at the loop head costs a vmcnt wait in front of every store)
    const float c2 = S.c2;
    // W1^T as the A operand: row = channel i, k = tap (0..8), 9 = bias (times a constant 1), 10..15 = 0
    h16x8 w1h, w1l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int tap = 8 * h + j;
        const float v = (tap < 9 ? w1[(tap < 9 ? tap : 0) * C1 + i] : tap == 9 ? b1[i] : 0.f) * S.sa2;   // conv1 writes conv2's scaled inputs
        const _Float16 vh = (_Float16)v;
        w1h[j] = vh;
        w1l[j] = (_Float16)(v - (float)vh);
    }
    // (fetched from LDS for every pixel block: eight registers that do not have to live through the matrix loop)
    unsigned char* const ldsW1 = ldsP + 2 * P12_BUF + lane * 16;
    if (wave == 0) {
        *reinterpret_cast<h16x8*>(ldsW1) = w1h;
        *reinterpret_cast<h16x8*>(ldsW1 + 64 * 16) = w1l;
    }

    // conv2: this lane's pixel (accumulator row i): image row 2 rp + (i >> 4), column 16 ch + (i & 15)
    const int row2 = 2 * rp + (i >> 4), col2 = 16 * ch + (i & 15);
    // conv1: pixel e of block b is patch pixel 32 b + i; lane (i, h) supplies taps 8 h + j: the lower half reads taps 0 .. 7 at
    // xin[(pr + dy) XW + pc + dx], the upper half tap 8 and then the table 1, 0, 0, ... through the same immediate offsets
    // Blocks by wave: the SIMD's issue arbitration favours its older wave (0-3), which finishes its matrix loop ~1000 cycles before
    // the younger one (4-7) when both carry the same load - so waves 0-3 take two blocks each (0 .. 7), waves 4-6 one (8 .. 10; wave 7 an empty one) and
    // waves 4-7 the LDS-DMA pieces.
    int xoff0[2], xoff1[2], poff[2];       // byte offsets: tap slot 0, tap slots 1..7 (from xin[0]), patch pixel
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int blk = wave < 4 ? wave + 4 * tt : wave + 4;
        const int e = 32 * blk + i, ec = min(e, P12_PIX - 1);
        const int pr = ec / PW, pc = ec - pr * PW;
        xoff0[tt] = (pr * XW + pc + (h ? 2 * XW + 2 : 0)) * 4;
        xoff1[tt] = h ? XIN_BYTES - 4 : (pr * XW + pc) * 4;           // (upper half: table[-1], so that slot j = 1 reads table[0])
        poff[tt] = e < P12_PIX ? c32_off(pr, pc, h) : -1;               // this lane's 16-byte chunk of the pixel in either plane
    }
    const int nblk = wave < 4 ? 2 : 1;        // (wave 7: a block of pixels beyond the patch - computed, never stored)
    // LDS-DMA pieces of this wave (wave-uniform)
    const int dma0 = wave >= 4 ? wave - 4 : -1, dma1 = (wave >= 4 && wave < 7) ? wave : -1;
    __syncthreads();

    // ---- the VALU / LDS work of an iteration, in pieces that ride in the gaps of the matrix loop ----
    // LDS-DMA piece m (0, 1) of the raw input of task t_(k+2): element e = (row r, column c) of the 12 x 36 region, a dword per lane
    // (per lane and piece ONE packed register, unpacked behind an opaque copy on every use: left to itself the compiler hoists the
    // row pointers of both pieces out of the tile loop - four registers it then spills and re-loads behind a vmcnt(0))
    unsigned dpk[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int piece = m == 0 ? dma0 : dma1;
        const int e = 64 * max(piece, 0) + lane;
        const int r = e / XW, c = e - r * XW, yi = r - 1;
        dpk[m] = (e < XIN && (unsigned)yi < (unsigned)Hin) ? (unsigned)(yi * Win + c - 1 + Win) | ((unsigned)c << 24) : 0x80000000u | (e < XIN ? 0u : 0x40000000u);
    }
    auto dma_piece = [&](int k, int m) {
        const int piece = m == 0 ? dma0 : dma1;
        if (piece < 0) return;                                 // wave-uniform
        unsigned win, xi0;
        tile_split(blockIdx.x + (unsigned)(k + 2) * G, xt, win, xi0);
        const int x0 = 32 * (int)xi0;
        const float* img = x + (size_t)win * (size_t)(Hin * Win);
        const unsigned xb = __builtin_amdgcn_readfirstlane(lds_address(lds) + (k & 1) * XIN_STRIDE + 256 * piece);
        unsigned q = dpk[m];
        asm volatile("" : "+v"(q));
        const bool ok = (int)q >= 0 && (unsigned)(x0 - 1 + (int)(q >> 24)) < (unsigned)Win;
        const float* src = ok ? img + ((int)(q & 0xffffffu) - Win + x0) : zeros;
        if (!(q & 0x40000000u)) dma4_to_lds(src, xb);
    };
    // conv1 of pixel block tt of task t_(k+1), stage c: 0 taps from LDS, 1 split, 2 the three MFMAs, 3 .. 6 ReLU + split + store of
    // channels 8 g + 4 h .. + 3 (g = c - 3)
    float xv[8];
    h16x8 xh, xl;
    f32x16 a1;
    u32x2 kh, kl;
    auto conv1_stage = [&](int k, int tt, int c) {
        const unsigned char* xin = lds + ((k + 1) & 1) * XIN_STRIDE;
        unsigned char* ph = ldsP + ((k + 1) & 1) * P12_BUF;
        if (c == 0) {
            const float* x0p = reinterpret_cast<const float*>(xin + xoff0[tt]);
            const float* x1p = reinterpret_cast<const float*>(xin + xoff1[tt]);
            xv[0] = x0p[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) xv[j] = x1p[(j / 3) * XW + (j % 3)];
        } else if (c == 1) {
            w1h = *reinterpret_cast<const h16x8*>(ldsW1);
            w1l = *reinterpret_cast<const h16x8*>(ldsW1 + 64 * 16);
            h16x4 h0, l0, h1, l1;
            split4(xv[0], xv[1], xv[2], xv[3], h0, l0);
            split4(xv[4], xv[5], xv[6], xv[7], h1, l1);
            xh = h16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            xl = h16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        } else if (c == 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) a1[q] = 0.f;
            a1 = MFMA16(w1l, xh, a1);
            a1 = MFMA16(w1h, xl, a1);
            a1 = MFMA16(w1h, xh, a1);
        } else {
            // quads g = 2 G, 2 G + 1 are the two 16-byte chunks of plane G, each half in one lane of the pair (i, h): after
            // v_permlane32_swap (upper lanes of the first operand <-> lower lanes of the second) lane (i, 0) holds chunk 0 and
            // lane (i, 1) chunk 1 whole - one 16-byte store per lane, plane and piece instead of two 8-byte stores with a
            // 32-byte lane stride (four-way bank conflicts: 37 % of the kernel's LDS cycles)
            const int g = c - 3;
            h16x4 vh, vl;
            split4(relu(a1[4 * g]), relu(a1[4 * g + 1]), relu(a1[4 * g + 2]), relu(a1[4 * g + 3]), vh, vl);
            const u32x2 bh = __builtin_bit_cast(u32x2, vh), bl = __builtin_bit_cast(u32x2, vl);
            if ((g & 1) == 0) {
                kh = bh;
                kl = bl;
            } else {
                const auto s0 = __builtin_amdgcn_permlane32_swap(kh.x, bh.x, false, false);   // {first operand, second operand} afterwards
                const auto s1 = __builtin_amdgcn_permlane32_swap(kh.y, bh.y, false, false);
                const auto s2 = __builtin_amdgcn_permlane32_swap(kl.x, bl.x, false, false);
                const auto s3 = __builtin_amdgcn_permlane32_swap(kl.y, bl.y, false, false);
                if (poff[tt] >= 0) {
                    const int off = (g >> 1) * P12_PLANE + poff[tt];
                    *reinterpret_cast<u32x4*>(ph + off) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(ph + P12_PIECE + off) = u32x4{s2[0], s3[0], s2[1], s3[1]};
                }
            }
        }
    };
    // conv2 + pool of this wave's 2 x 16 pixels of task t_k, `fill` riding in the matrix loop. The four pooled maxima stay in
    // registers (pm): bias, ReLU, split and the stores are deferred to store_pooled(), which rides in the NEXT matrix loop.
    float pm[4];
    auto conv2 = [&](int k, auto fill) {
        const unsigned char* ph = ldsP + (k & 1) * P12_BUF;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        conv_one_tile<3, P12_PLANE>(acc, wh, wl, ph, ph + P12_PIECE, [&](int dy, int dx) { return c32_off(row2 + dy, col2 + dx, h); }, fill);
        WS_STAMP(k, 1);
        // 2 x 2 pool inside the lane: registers q, q + 1 = columns 2 t, 2 t + 1 of the first row, q + 8, q + 9 of the second
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) pm[kk] = fmaxf(fmaxf(acc[2 * kk], acc[2 * kk + 1]), fmaxf(acc[2 * kk + 8], acc[2 * kk + 9]));
    };
    // outputs kk0, kk0 + 1 of task t_k. The output rows have a pitch of Wa = 16 x tiles pixels, so every lane's four pixels exist
    // (columns past the image land in the padding, which conv3's loader never reads): no bounds tests, one address, immediate offsets.
    auto store_pooled = [&](int k, int kk0) {
        unsigned win, xi0;
        tile_split(blockIdx.x + (unsigned)k * G, xt, win, xi0);
        const int x0 = 32 * (int)xi0 + 16 * ch;
        _Float16* o = out + ((size_t)(win * 4 + rp) * (size_t)Wa + (size_t)((x0 >> 1) + 2 * h)) * 64 + i;
        // (the pair kk0, kk0 + 1 split together; pixels 0, 1 / 4, 5 of the lane's four)
        typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
        unsigned ph2, pl2;
        split2(relu(fmaf(pm[kk0], c2, bias)), relu(fmaf(pm[kk0 + 1], c2, bias)), ph2, pl2);
        const h16x2 vh = __builtin_bit_cast(h16x2, ph2), vl = __builtin_bit_cast(h16x2, pl2);
        const int dpx = kk0 == 0 ? 0 : 4;
        o[dpx * 64] = vh[0];
        o[dpx * 64 + 32] = vl[0];
        o[(dpx + 1) * 64] = vh[1];
        o[(dpx + 1) * 64 + 32] = vl[1];
    };

    bool pending = false;                                      // pm holds the maxima of task t_(k-1), not stored yet
    for (int k = -2; k < nloc; ++k) {
        WS_STAMP(k, 0);
        if (k >= 0 && k + 2 < nloc) {
            // steady state: the LDS-DMA pieces (first: they have the whole loop to land), the conv1 stages and the stores of the
            // previous task sit behind the steps of the matrix loop
            // (the stores first: `s_waitcnt vmcnt(0)` in front of the barrier also waits for their write acknowledgements - behind
            // the last steps of the loop they cost every wave ~350 cycles there)
            if (nblk == 2) {
                conv2(k, [&](int f) {
                    if (f < 2) {
                        if (pending) store_pooled(k - 1, 2 * f);
                    } else if (f < 9) conv1_stage(k, 0, f - 2);
                    else if (f < 16) conv1_stage(k, 1, f - 9);
                });
            } else {
#ifndef F2_B_PRIO_STEPS
#define F2_B_PRIO_STEPS 0
#endif
                if (F2_B_PRIO_STEPS > 0) __builtin_amdgcn_s_setprio(1);
                conv2(k, [&](int f) {
                    if (f < 2) dma_piece(k, f);
                    else if (f < 16 && (f & 1) == 0) conv1_stage(k, 0, (f - 2) >> 1);
                    else if ((f == 3 || f == 5) && pending) store_pooled(k - 1, f - 3);
                    if (F2_B_PRIO_STEPS > 0 && f == F2_B_PRIO_STEPS - 1) __builtin_amdgcn_s_setprio(0);
                });
            }
        } else {
            // first and last iterations: what there is, one thing after the other
            if (pending) {
                store_pooled(k - 1, 0);
                store_pooled(k - 1, 2);
            }
            if (k + 2 < nloc) {
                dma_piece(k, 0);
                dma_piece(k, 1);
            }
            if (k + 1 >= 0 && k + 1 < nloc) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    if (tt < nblk) {
#pragma unroll
                        for (int c = 0; c < 7; ++c) conv1_stage(k, tt, c);
                    }
            }
            WS_STAMP(k, 4);
            if (k >= 0) conv2(k, [](int) {});
        }
        pending = k >= 0;
        if (k == nloc - 1 && pending) {
            store_pooled(k, 0);
            store_pooled(k, 2);
        }
        WS_STAMP(k, 2);
        dma_wait();
        WS_STAMP(k, 5);
        __syncthreads();
        WS_STAMP(k, 3);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3 + conv4 + pool
// ------------------------------------------------------------------------------------------------------------------
constexpr int PA_PIX = 6 * PW;                     // conv3 input patch: pooled conv2 rows -1 .. 4, columns c0 - 1 .. c0 + 32
constexpr int PA_PLANE = PA_PIX * 32;
constexpr int PA_PIECE = 2 * PA_PLANE;
constexpr int PA_BUF = 2 * PA_PIECE;
constexpr int PB_PIX = 4 * PW;                     // conv4 input patch: conv3 rows 0 .. 3, columns c0 .. c0 + 33
constexpr int PB_PLANE = PB_PIX * 64;
constexpr int PB_PIECE = 2 * PB_PLANE;
constexpr int PB_BUF = 2 * PB_PIECE;
constexpr int X_BUF = 2 * 8 * 64 * 16;             // partial sums of the second K half: 2 output tiles x 8 register quads x 64 lanes x 16 B
constexpr int PA_CHUNKS = 2 * PA_PIX * 4;          // 16-byte chunks of a patch A buffer
constexpr int PA_DMA = (PA_CHUNKS + 63) / 64;      // 1 KB LDS-DMA pieces of a patch A buffer
constexpr int PA_DMA_PER_WAVE = (PA_DMA + 3) / 4;  // 7
constexpr int BIAS3_BYTES = 2 * 2 * 16 * 4;        // conv3 biases in accumulator order: [output tile][lane half][register]
constexpr size_t LDS34 = 2 * (size_t)PA_BUF + 2 * (size_t)PB_BUF + 2 * (size_t)X_BUF + BIAS3_BYTES;
static_assert(LDS34 <= 160 * 1024, "one workgroup per CU");

__global__ __launch_bounds__(512) void k_conv34_ws(const uint4* __restrict__ in, const h16x8* __restrict__ w3s,
                                                   const float* __restrict__ b3, const h16x8* __restrict__ w4s,
                                                   const float* __restrict__ b4, float* __restrict__ out,
                                                   const uint4* __restrict__ zeros, int Win, int Wa, tile_div xt, unsigned ntile,
                                                   f2_split_scales S WS_STAMP_ARG) {
    // (b3 = conv3's biases x sa_3 sb_3: the accumulators start from them; b4 = conv4's x sa_dense1; out x sa_dense1)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 2 * PA_BUF;
    unsigned char* const ldsX = ldsB + 2 * PB_BUF;
    float* const ldsBias = reinterpret_cast<float*>(ldsX + 2 * X_BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int Wp = (Win - 2) / 2;                             // pooled conv4 width
    const unsigned G = gridDim.x;
    const int nloc = (int)((ntile - blockIdx.x + G - 1) / G); // tiles of this workgroup: blockIdx.x + j G

    // the columns of patch B that only conv4's discarded outputs read are never written: clear both buffers once
    for (int e = tid; e < (int)(2 * PB_BUF / 16); e += 512) reinterpret_cast<uint4*>(ldsB)[e] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 64) ldsBias[tid] = b3[(tid >> 5) * 32 + (tid & 3) + 8 * ((tid & 15) >> 2) + 4 * ((tid >> 4) & 1)];
    __syncthreads();

    if (wave < 4) {
        // ---------------- conv3: output tile nt (32 of the 64 channels), rows 2 rp, 2 rp + 1 ----------------
        const int nt = wave & 1, rp = wave >> 1;
        h16x8 wh[18], wl[18];
        {
            const h16x8* ph = w3s + h * C3 + nt * 32 + i;    // [piece][tap][kb][h][cout]
            const h16x8* pl = ph + 18 * 2 * C3;
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
        const f32x4* bias4 = reinterpret_cast<const f32x4*>(ldsBias + (nt * 2 + h) * 16);   // (16 registers too many to hold)
        const float c3 = S.c3;
        // LDS-DMA pieces of this wave, once: 16-byte slot s = 64 piece + lane of a patch A buffer = (piece, pixel, position) <-
        // chunk (position ^ swizzle(pixel)) of that pixel's piece (the XOR is its own inverse: the readers apply the same one).
        // rel = the chunk's index relative to (window, column c0 - 1) of the input; pcx = the pixel's patch column, or a value no
        // image is wide enough for where the patch row lies outside the image (rows -1 and 4) or the slot does not exist
        // (one register per piece: rel in bits 0 .. 23, pcx in bits 24 .. 31, 255 = never inside; f2_cnn_ws_supported bounds the width)
        unsigned relpc[PA_DMA_PER_WAVE];
#pragma unroll
        for (int m = 0; m < PA_DMA_PER_WAVE; ++m) {
            const int sl = 64 * (wave + 4 * m) + lane;
            // slot = (piece, plane kb, pixel, position): the 16 bytes of channels 16 kb + 8 (position ^ s(pixel)) of that pixel's piece
            const int piece = sl >= PA_PIX * 4 ? 1 : 0, rem = sl - piece * (PA_PIX * 4);
            const int kb = rem >= PA_PIX * 2 ? 1 : 0, rem2 = rem - kb * (PA_PIX * 2);
            const int pixel = rem2 >> 1, c = 2 * kb + ((rem2 & 1) ^ ((pixel >> 3) & 1));
            const int r = pixel / PW, pc = pixel - r * PW;
            const bool inside = sl < PA_CHUNKS && r >= 1 && r <= 4;
            relpc[m] = inside ? (unsigned)(((r - 1) * Wa + pc) * 8 + piece * 4 + c) | ((unsigned)pc << 24) : 0xFF000000u;
            asm volatile("" : "+v"(relpc[m]));
        }
        for (int j = -2; j <= nloc; ++j) {
            WS_STAMP(j, 0);
            // LDS-DMA piece m of tile j + 2: HBM -> the patch A buffer this wave's matrix loop read in the PREVIOUS iteration (this
            // iteration's loop reads the other one); pixels outside the image come from the zero buffer
            auto dma_piece = [&](int m) {
                const int dp = wave + 4 * m;                  // wave-uniform
                if (dp >= PA_DMA || j + 2 >= nloc) return;
                unsigned win, xi0;
                tile_split(blockIdx.x + (unsigned)(j + 2) * G, xt, win, xi0);
                const int cm1 = T34 * (int)xi0 - 1;           // image column of patch column 0
                const unsigned pa = __builtin_amdgcn_readfirstlane(lds_address(ldsA) + (j & 1) * PA_BUF + 1024 * dp);
                const uint4* img = in + ((size_t)win * 4 * (size_t)Wa + cm1) * 8;
                const uint4* src = (unsigned)(cm1 + (int)(relpc[m] >> 24)) < (unsigned)Win ? img + (relpc[m] & 0xFFFFFFu) : zeros;
                if (dp < PA_DMA - 1 || lane < PA_CHUNKS - 64 * (PA_DMA - 1)) dma16_to_lds(src, pa);
            };
            const int jt = j + 1;                             // conv3 works one tile ahead of conv4
            if (jt >= 0 && jt < nloc) {
                const unsigned char* pa = ldsA + (jt & 1) * PA_BUF;
                unsigned char* pbh = ldsB + (jt & 1) * PB_BUF;
                // the accumulators start from the biases (four 16-byte LDS reads per row instead of 32 additions at the end)
                f32x16 acc[2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bq = bias4[g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][4 * g + r] = bq[r];
                        acc[1][4 * g + r] = bq[r];
                    }
                }
                // ReLU + split + store of channels 32 nt + 8 g + 4 h .. + 3 of output row mt
                auto epilogue = [&](int mt, int g) {
                    const int pb = (2 * rp + mt) * PW + i;
                    const int base = b64_off(pb, 2 * nt) + 8 * h;   // chunk 2 nt + (g & 1) of plane g >> 1: base ^ ((g & 1) << 4)
                    h16x4 vh, vl;
                    // (scalar multiplies: a v_pk_mul_f32 per pair made this kernel 4 % slower, the multiplication folded into the
                    // v_fma_mix conversions changed nothing - profiles/r05_valu_diet.txt)
                    split4(relu(acc[mt][4 * g]) * c3, relu(acc[mt][4 * g + 1]) * c3, relu(acc[mt][4 * g + 2]) * c3,
                           relu(acc[mt][4 * g + 3]) * c3, vh, vl);
                    const int off = (g >> 1) * PB_PLANE + (base ^ ((g & 1) << 4));
                    *reinterpret_cast<h16x4*>(pbh + off) = vh;
                    *reinterpret_cast<h16x4*>(pbh + PB_PIECE + off) = vl;
                };
                PRIO_MATRIX();
                // behind the fragments of the matrix loop: the LDS-DMA pieces (first: they have the whole iteration to land) and,
                // once output row 0 is complete (fragment 17), its epilogue
                conv_two_rows<true, 3, PA_PLANE>(acc, wh, wl, pa, pa + PA_PIECE, [&](int pr, int dx) { return a32_off((2 * rp + pr) * PW + i + dx, h); },
                                                 [&](int f) {
                                                     if (f < PA_DMA_PER_WAVE) dma_piece(f);
                                                     else if (f >= 19 && f < 23) epilogue(0, f - 19);
                                                 });
                PRIO_VALU();
                WS_STAMP(j, 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) epilogue(1, g);
            } else {
#pragma unroll
                for (int m = 0; m < PA_DMA_PER_WAVE; ++m) dma_piece(m);
            }
            dma_wait();                                       // (issued a matrix loop and an epilogue ago)
            WS_STAMP(j, 2);
            __syncthreads();
            WS_STAMP(j, 3);
        }
    } else {
        // ---------------- loader + conv4: output tile nt, K half kh (input channels 32 kh .. 32 kh + 31) ----------------
        // (kh selects accumulator REGISTERS below - the half of them this wave finishes - so it has to be a compile-time constant:
        // as a wave-uniform run-time value every such access became an s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off triple, 64 moves
        // and 128 mode switches per tile; the role is instantiated for both halves and chosen by a wave-uniform branch)
        const int cw = wave - 4, nt = cw & 1;
        auto conv4_role = [&](auto KH) {
        constexpr int kh = decltype(KH)::value;
        h16x8 wh[18], wl[18];
        {
            const h16x8* ph = w4s + h * C4 + nt * 32 + i;    // [piece][tap][kb (4)][h][cout]
            const h16x8* pl = ph + 36 * 2 * C4;
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
        float bias = b4[nt * 32 + i];
        asm volatile("" : "+v"(bias));     // (settled here: see k_conv12_ws)
        const float c4 = S.c4;
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            acc[0][q] = 0.f;
            acc[1][q] = 0.f;
        }
        for (int j = -2; j <= nloc; ++j) {
            WS_STAMP(j, 0);
            // (1) the two K halves of tile j - 1 meet: each wave kept the accumulator registers of four of the eight pooled
            // outputs of a lane (first half: q = 0 .. 7, second half: q = 8 .. 15) and reads its partner's partial sums of those
            // from LDS; sum, pool, bias, ReLU, store
            if (j - 1 >= 0 && j - 1 < nloc) {
                unsigned win, xi0;
                tile_split(blockIdx.x + (unsigned)(j - 1) * G, xt, win, xi0);
                const int c0 = T34 * (int)xi0;
                const f32x4* xs = reinterpret_cast<const f32x4*>(ldsX + ((j - 1) & 1) * X_BUF) + ((nt * 2 + (kh ^ 1)) * 4) * 64 + lane;
                float* o = out + (size_t)win * (size_t)(Wp * C4) + nt * 32 + i;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {              // quad (mt, half): registers 8 kh + 4 (r4 & 1) .. + 3 of row mt = r4 >> 1
                    const f32x4 v = xs[r4 * 64];
                    const int mt = r4 >> 1, q = 8 * kh + 4 * (r4 & 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mt][q + r] += v[r];
                }
#pragma unroll
                for (int k = 4 * kh; k < 4 * kh + 4; ++k) {
                    const int q = 2 * k;
                    const int xl = (q & 3) + 8 * (q >> 2) + 4 * h;    // even conv4 column inside the tile
                    const int pxp = (c0 + xl) >> 1;
                    const float m = fmaxf(fmaxf(acc[0][q], acc[0][q + 1]), fmaxf(acc[1][q], acc[1][q + 1]));
                    if (xl < T34 && pxp < Wp) o[pxp * C4] = relu(fmaf(m, c4, bias));
                }
            }
            WS_STAMP(j, 4);
            WS_STAMP(j, 5);
            // (3) conv4 of tile j, this wave's K half
            if (j >= 0 && j < nloc) {
                const unsigned char* pbh = ldsB + (j & 1) * PB_BUF;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    acc[0][q] = 0.f;
                    acc[1][q] = 0.f;
                }
                PRIO_MATRIX();
                conv_two_rows<false, 3, PB_PLANE>(acc, wh, wl, pbh, pbh + PB_PIECE, [&](int pr, int dx) { return b64_off(pr * PW + i + dx, 2 * kh + h); },
                                                  [](int) {});
                PRIO_VALU();
                WS_STAMP(j, 1);
                // the registers the OTHER K half finishes: q = 8 (1 - kh) .. + 7 of both rows
                f32x4* xs = reinterpret_cast<f32x4*>(ldsX + (j & 1) * X_BUF) + ((nt * 2 + kh) * 4) * 64 + lane;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int mt = r4 >> 1, q = 8 * (kh ^ 1) + 4 * (r4 & 1);
                    xs[r4 * 64] = f32x4{acc[mt][q], acc[mt][q + 1], acc[mt][q + 2], acc[mt][q + 3]};
                }
            }
            WS_STAMP(j, 2);
            __syncthreads();
            WS_STAMP(j, 3);
        }
        };
        if (cw >> 1) conv4_role(std::integral_constant<int, 1>{});
        else conv4_role(std::integral_constant<int, 0>{});
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dense1 from pre-split activations
// ------------------------------------------------------------------------------------------------------------------
// out (n, D1) = ReLU(a (n, K) W (K, D1) + b). k_dense1_h16x3 (f2_cnn.hip) streams 2 KB of split weights per wave and K step for 6
// MFMAs (64 windows per workgroup) and lets the compiler place the waits of its prefetched loads: vmcnt(0) at the head of every
// chunk, the fragment requested a moment ago included. Here a weight fragment feeds 3 MT MFMAs (MT = 3: 96 windows), every
// global load of the K loop is issued and waited for by hand (two K steps ahead for the weights, a chunk ahead for the
// activations), and the activations' hi / lo chunks sit XOR-swizzled in LDS so that ds_read_b128 of 32 consecutive rows is
// conflict-free (chunk c of row r at r * 128 + ((c ^ ((r >> 1) & 7)) << 4): a 16-lane group of the read covers eight even and eight
// odd rows, whose halves of the 256-byte bank window are fixed by the row's parity). (conv4 leaving its outputs already split - two
// 2-byte stores or one packed dword per value - was measured too: it takes 18 us out of this kernel's staging and puts them into
// k_conv34_ws's combine, which runs beside matrix loops.)
// Workgroup shape (round 5, profiles/r05_dense1_knockouts.txt): with two resident 6-wave workgroups per CU and every other cost
// knocked out the kernel took 98 of its 122 us per 14 240 windows - the matrix time of a SIMD that runs FOUR of the CU's twelve
// waves, and the same with one workgroup per CU in two rounds: the times fit both workgroups' waves going 2, 2, 1, 1 to the
// SIMDs (the two-wave workgroups of the filterbank kernel do spread: 4 waves per workgroup change nothing there). Now 12 waves =
// 2 row halves x 6 output tiles: three waves per SIMD by construction, one workgroup per CU, 164 registers. What is left (113 us against 58 us of matrix time): the activation fragments -
// 2 KB of LDS reads per 3 MT MFMAs and wave, 22 us with everything else knocked out - and the launch's fixed costs.
constexpr int D1W_WAVES = 6, D1W_KC = 64, D1W_TILES = 17, D1W_NPAD = D1W_TILES * 32, D1W_N = 516;
constexpr int D1W_GROUPS = (D1W_TILES + D1W_WAVES - 1) / D1W_WAVES;
#ifndef F2_D1W_HALVES
#define F2_D1W_HALVES 2
#endif
constexpr int D1W_HALVES = F2_D1W_HALVES;      // row halves of a workgroup: 6 x 2 = 12 waves, three on every SIMD
constexpr int D1W_THREADS = D1W_WAVES * D1W_HALVES * 64;
template <int MT>
__global__ __launch_bounds__(D1W_THREADS, 3) void k_dense1_ws(const float* __restrict__ a, const h16x8* __restrict__ ws,
                                                                 const float* __restrict__ bias, float* __restrict__ out, int K,
                                                                 int64_t n, f2_split_scales S WS_STAMP_ARG) {
    constexpr int ROWS = 32 * MT * D1W_HALVES, PIECE = ROWS * 128, BUF = 2 * PIECE;
    static_assert(2 * BUF <= 160 * 1024 && PIECE + 32 * MT * 128 < 65536, "LDS of a CU; immediate ds offsets from a half's base");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    // Workgroup -> (row block rb, tile group ty). The D1W_GROUPS workgroups that share a row block's activations sit on the same
    // XCD (workgroup L runs on XCD L % 8) and next to each other in dispatch order, so that the block's K chunks come from HBM
    // once and from that XCD's L2 for the others; with gridDim = (row blocks, groups) they were a whole grid row apart and
    // every group re-read all activations from HBM (3 x 875 MB per 113 920 windows: the kernel's bound).
    const unsigned gq = blockIdx.x / (8 * D1W_GROUPS), gr = blockIdx.x - gq * (8 * D1W_GROUPS);
    const int ty = gr >> 3;
    const int64_t rb = (int64_t)gq * 8 + (gr & 7);
    const int64_t w0 = rb * ROWS;
    if (w0 >= n) return;                                             // (workgroup-uniform, ahead of every barrier)
    const int nchunks = K / D1W_KC, nsteps = nchunks * 4;
    const int hf = wave / D1W_WAVES;                                 // row half: rows 32 MT hf .. of the workgroup's
    const int nt = ty * D1W_WAVES + (wave - hf * D1W_WAVES);
    const bool live = nt < D1W_TILES;
    const int ntc = live ? nt : D1W_TILES - 1;                       // (idle waves load a valid tile and discard it)
    const h16x8* wh = ws + (int64_t)h * D1W_NPAD + ntc * 32 + i;    // w[piece][chunk][ks][h][n (544)][8]
    const h16x8* wl = wh + (int64_t)nchunks * 4 * 2 * D1W_NPAD;

    // Staging: thread slot s = tid + 384 m = (row, chunk c of eight K values): two 16-byte loads (eight float32), split, two
    // 16-byte LDS stores (hi and lo chunk) at position c ^ ((row >> 1) & 7) of the row in either piece; rows
    // past n read row n - 1 (their results are not stored). Through registers rather than by LDS-DMA: vmcnt counts in order, so a
    // wait for a weight fragment also waits for everything issued before it - either way two K steps after its issue.
    constexpr int NPAIR = ROWS * 8, PERT = (NPAIR + D1W_THREADS - 1) / D1W_THREADS;
    unsigned srcoff[PERT];
    int dstoff[PERT];
#pragma unroll
    for (int m = 0; m < PERT; ++m) {
        const int sl = min((int)threadIdx.x + D1W_THREADS * m, NPAIR - 1);
        const int row = sl >> 3, c = sl & 7;
        const int64_t wr = w0 + row < n ? w0 + row : n - 1;
        srcoff[m] = (unsigned)(wr * (int64_t)K + c * 8) * 4u;                    // bytes (n x K dwords < 4 GB: f2_launch_dense1_ws checks)
        dstoff[m] = (int)threadIdx.x + D1W_THREADS * m < NPAIR ? row * 128 + ((c ^ ((row >> 1) & 7)) << 4) : -1;
    }
    // Every global load of the K loop is issued through inline asm and waited for by hand: left to the compiler, the waits at the
    // head of the loop come out as vmcnt(0) - the fragment requested a moment ago included (a full L2 latency per chunk).
    // vmcnt counts in order: a wait for N = everything but the N youngest loads has arrived.
    u32x4 ar[2 * PERT];
    auto gload = [](u32x4& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory"); };
    auto load_a = [&](int kc) {
#ifdef F2_D1W_KO_A      // timing knock-out: the activations of chunk 0 again (cache hits)
        kc = 0;
#endif
#pragma unroll
        for (int m = 0; m < PERT; ++m) {
            const unsigned char* p = reinterpret_cast<const unsigned char*>(a) + srcoff[m] + kc * (D1W_KC * 4);
            gload(ar[2 * m], p);
            gload(ar[2 * m + 1], p + 16);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int m = 0; m < PERT; ++m) {
            const f32x4 x = __builtin_bit_cast(f32x4, ar[2 * m]), y = __builtin_bit_cast(f32x4, ar[2 * m + 1]);
            h16x4 h0, l0, h1, l1;
            split4(x[0], x[1], x[2], x[3], h0, l0);
            split4(y[0], y[1], y[2], y[3], h1, l1);
            if (dstoff[m] >= 0) {
                *reinterpret_cast<h16x8*>(lds + buf * BUF + dstoff[m]) = h16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                *reinterpret_cast<h16x8*>(lds + buf * BUF + PIECE + dstoff[m]) = h16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
        }
    };
    // this lane's 16 bytes of K step ks in row i of an M tile: + 4096 t + PIECE piece + BUF buffer (immediates)
    int aoff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) aoff[ks] = i * 128 + (((2 * ks + h) ^ ((i >> 1) & 7)) << 4);

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    // weight fragments: a ring of four K steps (slot = ks, compile-time)
    u32x4 bh[4], bl[4];
    auto load_b = [&](int step, int slot) {
#ifdef F2_D1W_KO_W      // timing knock-out (tools/build_variant.sh): every step re-reads the fragment of step `slot` (L1 hits)
        step = slot;
#endif
        gload(bh[slot], wh + (int64_t)step * 2 * D1W_NPAD);
        gload(bl[slot], wl + (int64_t)step * 2 * D1W_NPAD);
    };
    // (the fragment is an operand of the wait, so that its MFMAs cannot be scheduled ahead of it)
#define D1W_WAIT(N, slot) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bh[slot]), "+v"(bl[slot])::"memory")
    // Activation fragments of a K step: MT x (hi, lo) ds_read_b128, requested a whole step ahead of the MFMAs that use them
    // (two register sets). Left to the compiler they are read two at a time right in front of their MFMAs - three exposed LDS
    // latencies per step and wave.
    h16x8 fah[2][MT], fal[2][MT];
    auto fetch = [&](const unsigned char* pa, int ks, int set) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#ifdef F2_D1W_KO_LDSR   // timing knock-out: no LDS reads in the loop
            fal[set][t] = __builtin_bit_cast(h16x8, bh[t]);
            fah[set][t] = __builtin_bit_cast(h16x8, bl[t]);
#else
            fal[set][t] = *reinterpret_cast<const h16x8*>(pa + aoff[ks] + t * 4096 + PIECE);
            fah[set][t] = *reinterpret_cast<const h16x8*>(pa + aoff[ks] + t * 4096);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // The three products of an accumulator depend on each other, so the M tiles take turns: product p of tiles 0 .. MT - 1,
    // then product p + 1 (per accumulator the order of summation is lo x hi, hi x lo, hi x hi as everywhere).
    auto step_mfmas = [&](int ks, int set) {
        const h16x8 wb_h = __builtin_bit_cast(h16x8, bh[ks]), wb_l = __builtin_bit_cast(h16x8, bl[ks]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = MFMA16(fal[set][t], wb_h, acc[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = MFMA16(fah[set][t], wb_l, acc[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = MFMA16(fah[set][t], wb_h, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // Fragments are requested THREE steps ahead (the ring's fourth slot is the one the previous step has just consumed). Issue
    // order of a chunk: activations of chunk kc + 1 (2 PERT loads), then per step one fragment pair; a wait for the fragment
    // of steps 0 - 2 leaves the NA = 2 PERT + 6 younger loads in flight, step 3's the six of the three pairs behind it -
    // which also means the activations, older than that fragment, have arrived.
    load_a(0);
    load_b(0, 0);
    load_b(1, 1);
    load_b(2, 2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ar[0]), "+v"(ar[2 * PERT - 1])::"memory");
#pragma unroll
    for (int m = 1; m + 1 < 2 * PERT; ++m) asm volatile("" : "+v"(ar[m]));
    store_a(0);
    __syncthreads();
    constexpr int NA = 2 * PERT + 6;
    static_assert(NA == 10 || NA == 12, "MT = 3 or 4");
#define D1W_WAIT_NA(slot)                      \
    do {                                       \
        if constexpr (NA == 10) D1W_WAIT(10, slot); \
        else D1W_WAIT(12, slot);               \
    } while (0)
    int kc = 0;
    for (; kc + 1 < nchunks; ++kc) {
        const unsigned char* pa = lds + (kc & 1) * BUF + hf * (32 * MT * 128);
        WS_STAMP(kc, 0);
        fetch(pa, 0, 0);
        load_a(kc + 1);
        load_b(kc * 4 + 3, 3);
        fetch(pa, 1, 1);
        D1W_WAIT_NA(0);
        WS_STAMP(kc, 1);
        step_mfmas(0, 0);
        load_b(kc * 4 + 4, 0);
        fetch(pa, 2, 0);
        D1W_WAIT_NA(1);
        step_mfmas(1, 1);
        load_b(kc * 4 + 5, 1);
        fetch(pa, 3, 1);
        D1W_WAIT_NA(2);
        step_mfmas(2, 0);
        load_b(kc * 4 + 6, 2);
        D1W_WAIT(6, 3);
#pragma unroll
        for (int m = 0; m < 2 * PERT; ++m) asm volatile("" : "+v"(ar[m]));
        step_mfmas(3, 1);
        WS_STAMP(kc, 2);
#ifndef F2_D1W_KO_STORE  // timing knock-out: no split, no LDS stores
        store_a((kc + 1) & 1);                                       // (that buffer was read in chunk kc - 1: behind that chunk's barrier)
#endif
        WS_STAMP(kc, 4);
#ifndef F2_D1W_KO_BAR    // timing knock-out: no barrier
        __syncthreads();
#endif
        WS_STAMP(kc, 3);
    }
    {
        // last chunk: fragments 0 - 2 are in flight, 3 still to request
        const unsigned char* pa = lds + (kc & 1) * BUF + hf * (32 * MT * 128);
        fetch(pa, 0, 0);
        load_b(kc * 4 + 3, 3);
        fetch(pa, 1, 1);
        D1W_WAIT(6, 0);
        step_mfmas(0, 0);
        fetch(pa, 2, 0);
        D1W_WAIT(4, 1);
        step_mfmas(1, 1);
        fetch(pa, 3, 1);
        D1W_WAIT(2, 2);
        step_mfmas(2, 0);
        D1W_WAIT(0, 3);
        step_mfmas(3, 1);
    }
#undef D1W_WAIT_NA
#undef D1W_WAIT
    const int col = nt * 32 + i;
    if (live && col < D1W_N) {
        float b = bias[col];
        asm volatile("" : "+v"(b));        // (settled here: see k_conv12_ws)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t wr = w0 + hf * (32 * MT) + t * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (wr < n) out[wr * D1W_N + col] = relu(fmaf(acc[t][q], S.cd, b));
            }
    }
}

}  // namespace

// conv1 .. conv4 + pools of n windows: x (n, H1, W1) float32 -> a2s (n, 4, W1/2 - 1, [hi 32 | lo 32]) fp16, x sa_3 (scratch) ->
// a4 (n, 1, Wp2, 64) float32. Only for windows with four pooled rows after conv2 (f2_cnn_ws_supported).
bool f2_cnn_ws_supported(int rows, int channels) {
    const int Hp1 = (rows - 2) / 2, Wp1 = (channels - 2) / 2;
    // Wp1 <= 200: k_conv34_ws packs a patch column and an image offset into one register; Wp1 >= 8: the padded conv2 output
    // (rows of 16 x tiles pixels, 128 bytes each) fits in the space f2_cnn_workspace_floats reserves for conv2's and conv3's
    return Hp1 == 4 && Wp1 >= 8 && Wp1 <= 200;
}

#ifdef F2_WS_STAMPS
static void ws_stamp_report(f2_ctx* ctx, const char* name, unsigned long long* d_stamps) {
    constexpr size_t N = 8 * 8 * 32 * 8;
    std::vector<unsigned long long> hst(N);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return;
    if (hipMemcpy(hst.data(), d_stamps, sizeof(unsigned long long) * N, hipMemcpyDeviceToHost) != hipSuccess) return;
    fprintf(stderr, "[ws stamps] %s: mean cycles per iteration (s_memtime), iterations 8..39 of workgroups 0..7\n", name);
    for (int wave = 0; wave < 8; ++wave) {
        double d[8] = {0}, iter = 0, clk = 0;
        int cnt = 0, ccnt = 0;
        for (int wg = 0; wg < 8; ++wg)
            for (int it = 0; it < 32; ++it) {
                const unsigned long long* sp = hst.data() + (((size_t)wg * 8 + wave) * 32 + it) * 8;
                if (sp[0] == 0 || sp[3] == 0) continue;
                for (int sl = 1; sl < 6; ++sl)
                    if (sp[sl]) d[sl] += (double)(sp[sl] - sp[0]);
                ++cnt;
                if (it + 1 < 32) {
                    const unsigned long long* sn = sp + 8;
                    if (sn[0] && sn[7] > sp[7]) {
                        iter += (double)(sn[0] - sp[0]);
                        clk += (double)(sn[0] - sp[0]) / (double)(sn[7] - sp[7]) * 100e6;
                        ++ccnt;
                    }
                }
            }
        if (!cnt) continue;
        fprintf(stderr, "  wave %d: +matrix loop %.0f  +epilogue %.0f  +barrier %.0f  (slot4 %.0f slot5 %.0f)  iteration %.0f cycles, clock %.2f GHz\n",
                wave, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt, ccnt ? iter / ccnt : 0.0, ccnt ? clk / ccnt / 1e9 : 0.0);
    }
}
#endif

int f2_launch_cnn_ws(f2_ctx* ctx, const f2_cnn* cnn, const float* d_x, int64_t n, void* a2s, float* a4) {
#ifdef F2_WS_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) F2_HIP(ctx, hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 8 * 8 * 32 * 8));
    F2_HIP(ctx, hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * 8 * 8 * 32 * 8, ctx->stream));
#define WS_STAMP_PASS , d_stamps
#else
#define WS_STAMP_PASS
#endif
    const int H1 = cnn->rows, W1 = cnn->channels;
    const int Wo = W1 - 2, Wp1 = Wo / 2, Wp2 = (Wp1 - 2) / 2;
    const int grid_max = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const int Wa = 16 * (((Wo / 2) * 2 + 31) / 32);           // row pitch of the conv2 output in pixels (f2_cnn_ws_a2_floats)
    {
        const int xtiles = ((Wo / 2) * 2 + 31) / 32;
        const int64_t ntask = n * xtiles;
        F2_CHECK(ctx, ntask < (int64_t(1) << 27), F2_ERR_UNSUPPORTED, "CNN chunk too large");
        const tile_div xt = {(unsigned)xtiles, (unsigned)(((uint64_t(1) << 32) + xtiles - 1) / xtiles)};
        static_assert(LDS12 <= 160 * 1024, "one workgroup per CU");
        F2_HIP(ctx, hipFuncSetAttribute((const void*)k_conv12_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS12));
        const unsigned grid = (unsigned)(ntask < grid_max ? ntask : grid_max);
        hipLaunchKernelGGL(k_conv12_ws, dim3(grid), dim3(512), LDS12, ctx->stream, d_x, cnn->t(0), cnn->t(1),
                           (const h16x8*)(cnn->blob16 + cnn->off16[0]), cnn->sbias + F2_SB_B2, (_Float16*)a2s, (const float*)cnn->zeros, H1, W1,
                           Wa, xt, (unsigned)ntask, cnn->sc WS_STAMP_PASS);
        F2_HIP(ctx, hipGetLastError());
#ifdef F2_WS_STAMPS
        ws_stamp_report(ctx, "k_conv12_ws (slot4 / slot5 = VALU phase of waves 4-7 / 0-3 done)", d_stamps);
        F2_HIP(ctx, hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * 8 * 8 * 32 * 8, ctx->stream));
#endif
    }
    {
        const int xtiles = (2 * Wp2 + T34 - 1) / T34;
        const int64_t ntile = n * xtiles;
        F2_CHECK(ctx, ntile < (int64_t(1) << 27), F2_ERR_UNSUPPORTED, "CNN chunk too large");
        const tile_div xt = {(unsigned)xtiles, (unsigned)(((uint64_t(1) << 32) + xtiles - 1) / xtiles)};
        F2_HIP(ctx, hipFuncSetAttribute((const void*)k_conv34_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS34));
        const unsigned grid = (unsigned)(ntile < grid_max ? ntile : grid_max);
        hipLaunchKernelGGL(k_conv34_ws, dim3(grid), dim3(512), LDS34, ctx->stream, (const uint4*)a2s,
                           (const h16x8*)(cnn->blob16 + cnn->off16[1]), cnn->sbias + F2_SB_B3I, (const h16x8*)(cnn->blob16 + cnn->off16[2]),
                           cnn->sbias + F2_SB_B4, a4, (const uint4*)cnn->zeros, Wp1, Wa, xt, (unsigned)ntile, cnn->sc WS_STAMP_PASS);
        F2_HIP(ctx, hipGetLastError());
#ifdef F2_WS_STAMPS
        ws_stamp_report(ctx, "k_conv34_ws (waves 0-3 conv3; 4-7 conv4: slot4 = K halves combined + stored, slot5 = LDS-DMA issued)", d_stamps);
#endif
    }
    return F2_OK;
}

// dense1 of n windows: a4 (n, K) float32 -> a5 (n, 516) float32
int f2_launch_dense1_ws(f2_ctx* ctx, const f2_cnn* cnn, const float* a4, int64_t n, int K, float* a5) {
    F2_CHECK(ctx, K % D1W_KC == 0 && K >= 2 * D1W_KC && n * (int64_t)K * 4 < (int64_t(1) << 32), F2_ERR_UNSUPPORTED,
             "dense1: %lld windows x %d inputs", (long long)n, K);
#ifndef F2_D1W_MT
#define F2_D1W_MT 3
#endif
    constexpr int MT = F2_D1W_MT;
#ifdef F2_D1W_ONE_WG     // timing experiment: one workgroup per CU
    constexpr int LDSB = 100 * 1024;
#else
    constexpr int LDSB = 2 * 2 * 32 * MT * D1W_HALVES * 128;
#endif
    F2_HIP(ctx, hipFuncSetAttribute((const void*)k_dense1_ws<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    const int64_t nrb = (n + 32 * MT * D1W_HALVES - 1) / (32 * MT * D1W_HALVES);
    const dim3 grid((unsigned)((nrb + 7) / 8 * 8 * D1W_GROUPS));
#ifdef F2_WS_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) F2_HIP(ctx, hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 8 * 8 * 32 * 8));
    F2_HIP(ctx, hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * 8 * 8 * 32 * 8, ctx->stream));
#endif
    hipLaunchKernelGGL(k_dense1_ws<MT>, grid, dim3(D1W_THREADS), LDSB, ctx->stream, a4,
                       (const h16x8*)(cnn->blob16 + cnn->off16[3]), cnn->t(9), a5, K, n, cnn->sc WS_STAMP_PASS);
    F2_HIP(ctx, hipGetLastError());
#ifdef F2_WS_STAMPS
    ws_stamp_report(ctx, "k_dense1_ws (waves 0-7 of 12; chunk = iteration: +matrix loop = wait for the first weight fragment, +epilogue = four steps issued, slot4 = split + stores done, +barrier = through the barrier)", d_stamps);
#endif
    return F2_OK;
}
