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
// TB-sample block of results is written to an LDS tile (one row per lane) and then streamed out row by row, so every
// global store instruction covers 64/TB row segments of 128 contiguous bytes.
//
// Those segments must also START on a 128-byte line, or every store leaves two partly written lines behind that
// the next tile of the same row completes microseconds later - by then evicted from L2 (8 GB stream through it),
// and a partial write costs HBM a read-modify-write: measured 2.3x on the whole kernel for rows that are not
// aligned (n = 15999 or 16002 against 16000; 64-byte alignment still costs 15-50 %) - and real utterance lengths
// are arbitrary. So each row is stored with its own lag d = (element index of its first sample) mod TB: after
// tile [t0, t0+TB) the samples [t0 - d, t0 - d + TB) of the row go out, one aligned 128-byte line. In LDS a row
// has 2 TB columns; a lane writes its tile to columns [d, d + TB), keeps it in registers as well and, after the
// store phase, copies it to columns (d + j + TB) mod 2 TB - which puts the last d samples at [0, d), in front of
// the next tile (the others land in the unused columns behind it) - through addresses computed once per
// utterance. The store phase therefore reads the fixed columns [0, TB) of every row, and the rows of one store
// instruction (q, q + RQ, ...) share their lag: every address of the store phase is wave-uniform base + fixed lane
// offset.
//
// Bound: the float64 FMA pipe (13 ops per sample-channel; 17 in the general form); HBM traffic is 8*C*N bytes
// written per utterance (4*C*N as the float32 hand-off to K2) + 2*N read.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "f2_internal.h"

namespace {

template <typename OutT>
struct TileGeo {
    static constexpr int TB = 128 / (int)sizeof(OutT);       // samples per tile = one 128-byte line of a row (lag < TB)
    static constexpr int RPS = 64 / TB;                      // rows per store instruction
    static constexpr int RQ = 64 / RPS;                      // store instructions per tile; RQ * pitch % TB == 0
    static constexpr int W = 2 * TB;                         // columns of a row in LDS
    // Row pitch in LDS, chosen per utterance: lane l writes column d_l + s of row l, d_l = (phase + l * pstep) mod TB, so
    // the 32 lanes of a write group land on banks l * (pitch + pstep) + const: conflict-free when pitch + pstep is odd
    // (and up to 32-way when it is a multiple of 32). pstep even -> W + 1, pstep odd -> W + 2.
    static constexpr int PITCH0 = W + 1;
    static constexpr int PITCH_MAX = W + 2;
};
constexpr int SEG_ALIGN = 32;   // host: segment lengths of the time-split path, a multiple of every tile size
// Wavefronts per workgroup. The waves of a workgroup are independent (own utterance/channel group, own LDS tile, no
// workgroup barrier); they only share a workgroup so that the dispatcher places them one per SIMD of a CU. With
// one-wave workgroups the 2000 waves of the benchmark batch land unevenly (some SIMDs run three, others one) and the
// kernel takes as long as the fullest SIMD.
#ifndef F2_K1_WAVES_F32
#define F2_K1_WAVES_F32 4     // float32 hand-off tiles: 4 x 16.9 KB of LDS
#endif
#ifndef F2_K1_WAVES_F64
#define F2_K1_WAVES_F64 2     // float64 output tiles: 2 x 17.4 KB (the store-bound variant gains nothing beyond two)
#endif
// (MODE 2 of the time-split path keeps a 32 KB matrix table per wave in LDS: two waves per workgroup)
template <typename OutT, int MODE = 0>
constexpr int waves_per_block() { return MODE == 2 ? 2 : sizeof(OutT) == 4 ? F2_K1_WAVES_F32 : F2_K1_WAVES_F64; }

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
// Time-split execution for small batches (MODE 1 / 2; MODE 0 is the plain kernel). A wave that walks a whole
// utterance needs ~47 ns per sample whatever the batch size, so B utterances never take less than N * 47 ns and a
// small batch leaves the chip idle. The recurrences are linear: with S the eight state words entering a segment of
// L samples and E the state a run from ZERO state leaves at its end, the true state at the end is  M S + E,
// M = T^L (T = the 8 x 8 zero-input transition of one sample; per channel, from the host in long double). So
//   pass 1 (MODE 1): every (unit, segment) wave runs its segment from zero state, stores nothing but E;
//   pass 2 (MODE 2): every (unit, segment) wave first chains S_j = M S_(j-1) + E_(j-1) over the segments before
//                    its own (at most K-1 steps of 64 FMAs, M in LDS), then runs its segment from S_j and stores.
// Twice the arithmetic, 1/K of the latency (+ the chain); the results differ from the serial run by float64 rounding.
struct SplitArgs {
    int K;                  // segments per utterance
    int L;                  // samples per segment (multiple of SEG_ALIGN)
    double* states;         // [unit][K][8][64 lanes]: E of every segment
    const double* mtab;     // [C][8][8]: M = T^L, row major
    const int* order;       // MODE 3: units in the order they are handed out (longest first)
    int* queue;             // MODE 3: next position in `order` (zeroed before the launch)
    const int* uflag;       // per utterance, or NULL: utterances whose flag is 0 are skipped (served by the spectral kernel)
    int qwaves;             // MODE 3: waves of the launch (0 = from the batch)
};

// One unit of work = (utterance, group of 64 channels[, segment]) on one wave.
template <typename WaveT, typename OutT, bool A2ZERO, int MODE>
__device__ __forceinline__ void filterbank_unit(const WaveT* __restrict__ wave, const int64_t* __restrict__ offsets,
                                                const double* __restrict__ coefs, int C, int groups,
                                                double* __restrict__ out, float* __restrict__ alt,
                                                const int64_t* __restrict__ alt_off, const SplitArgs& sp, int unit, int seg,
                                                int lane, OutT* tile, double* xs,
                                                double (*mshw)[64]) {
    using G = TileGeo<OutT>;
    constexpr int TB = G::TB, RQ = G::RQ;
    const int K = (MODE == 1 || MODE == 2) ? sp.K : 1;
    const int b = unit / groups;
    const int c0 = (unit % groups) * 64;
    if (sp.uflag && !sp.uflag[b]) return;
    const int64_t off = offsets[b];
    const int64_t N = offsets[b + 1] - off;
    if (N <= 0) return;
    const int64_t t_begin = (MODE == 1 || MODE == 2) ? (int64_t)seg * sp.L : 0;
    const int64_t t_end = (MODE == 1 || MODE == 2) ? min(N, t_begin + sp.L) : N;
    if (t_begin >= N) return;

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
    if constexpr (MODE == 2) {
        if (seg > 0) {
            // true state entering this segment: S_j = M S_(j-1) + E_(j-1), S_0 = 0
            const double* mrow = sp.mtab + (size_t)c * 64;
#pragma unroll 8
            for (int i = 0; i < 64; ++i) mshw[i][lane] = mrow[i];
            wave_sync();
            double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const double* e = sp.states + ((size_t)unit * K) * 8 * 64 + lane;
            // The end states E_j of the segments before this one, four steps ahead of the chain that consumes them: read one
            // step at a time, every step waited out a round trip to L2 for its eight words (31 steps: ~40 of the 61 us a single
            // 1 s utterance spent in this kernel).
            constexpr int AHEAD = 4;
            double eb[AHEAD][8];
#pragma unroll
            for (int a = 0; a < AHEAD; ++a)
#pragma unroll
                for (int r = 0; r < 8; ++r) eb[a][r] = e[(size_t)min(a, K - 1) * (8 * 64) + r * 64];
            for (int j0 = 0; j0 < seg; j0 += AHEAD) {
#pragma unroll
                for (int a = 0; a < AHEAD; ++a) {
                    if (j0 + a < seg) {                       // (wave-uniform)
                        double Sn[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            double acc = eb[a][r];
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc = fma(mshw[r * 8 + q][lane], S[q], acc);
                            Sn[r] = acc;
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) S[r] = Sn[r];
                        const int jn = min(j0 + a + AHEAD, K - 1);
#pragma unroll
                        for (int r = 0; r < 8; ++r) eb[a][r] = e[(size_t)jn * (8 * 64) + r * 64];
                    }
                }
            }
            z10 = S[0], z11 = S[1], z20 = S[2], z21 = S[3], z30 = S[4], z31 = S[5], z40 = S[6], z41 = S[7];
        }
    }

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
    // row pitch in elements, 128-byte phase of row c0, and this lane's own row: its tile goes to columns [d_lane, d_lane + TB)
    const int64_t pitch = N * ROWMUL;
    const int pstep = (int)(pitch & (TB - 1));
    const int phase_c0 = (int)((reinterpret_cast<uintptr_t>(obase) / sizeof(OutT) + (uint64_t)c0 * (uint64_t)pitch) & (TB - 1));
    const int d_lane = (phase_c0 + lane * pstep) & (TB - 1);
    const bool lagged = (pstep | phase_c0) != 0;              // wave-uniform; false: every row starts on a line, d = 0
    const int lpitch = G::PITCH0 + (pstep & 1);                // LDS row pitch of this utterance (TileGeo)
    OutT* const wb = tile + lane * lpitch + d_lane;
    const OutT* const rb = tile + srow * RQ * lpitch + scol;   // row q + srow*RQ of store q: rb + q*lpitch
    [[maybe_unused]] OutT* carry[TB];                         // where sample j of a tile waits for the next window
    if constexpr (MODE != 1) {
#pragma unroll
        for (int j = 0; j < TB; ++j) carry[j] = tile + lane * lpitch + ((d_lane + j + TB) & (2 * TB - 1));
    }
    OutT* const orow = reinterpret_cast<OutT*>(obase) + (size_t)c0 * (size_t)pitch;
    const uint32_t voff = (uint32_t)(((size_t)(srow * RQ) * (size_t)pitch + (size_t)scol) * sizeof(OutT));   // (fits32)
    // Samples [t0 - d, t0 - d + TB) of every row, d the row's lag; `fast`: all of them inside [t_begin, t_end), all 64
    // rows exist, offsets fit 32 bits - wave-uniform row base + fixed lane offset, no checks.
    auto store_window_p = [&](int64_t t0, bool fast, auto lp) {
        constexpr int PITCH = decltype(lp)::value;
        if (fast) {
            OutT vals[RQ];   // all LDS reads first (distinct registers), then the stores: no load-use wait per row
#pragma unroll
            for (int q = 0; q < RQ; ++q) vals[q] = rb[q * PITCH];
            if (!lagged) {
                // every row on a line: one lane pointer stepped by the row pitch
                char* p = reinterpret_cast<char*>(orow + t0) + voff;
                const int64_t rstep = pitch * (int64_t)sizeof(OutT);
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    *reinterpret_cast<OutT*>(p) = vals[q];
                    p += rstep;
                }
            } else {
                // row q starts dq samples early: wave-uniform row pointer (scalar unit) + the fixed lane offset
                const char* rowp = reinterpret_cast<const char*>(orow + t0);
                const int64_t rstep = pitch * (int64_t)sizeof(OutT);
                int dq = phase_c0;
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    *reinterpret_cast<OutT*>(const_cast<char*>(rowp) - dq * (int)sizeof(OutT) + voff) = vals[q];
                    rowp += rstep;
                    dq = (dq + pstep) & (TB - 1);
                }
            }
        } else {
            int dq = phase_c0;
#pragma unroll 8
            for (int q = 0; q < RQ; ++q) {
                const int row = q + srow * RQ;
                const int64_t t = t0 - dq + scol;
                if (t >= t_begin && t < t_end && c0 + row < C) orow[(size_t)row * (size_t)pitch + (size_t)t] = rb[q * PITCH];
                dq = (dq + pstep) & (TB - 1);
            }
        }
    };
    auto store_window = [&](int64_t t0, bool fast) {
        if (pstep & 1) store_window_p(t0, fast, std::integral_constant<int, G::PITCH0 + 1>{});
        else store_window_p(t0, fast, std::integral_constant<int, G::PITCH0>{});
    };

    // The input block is fetched one block ahead: vmcnt retires in issue order (stores included), so a load
    // issued after a block's 32 row stores would make the wave wait for those stores every block.
    WaveT xnext = (lane < TB && t_begin + lane < N) ? w[t_begin + lane] : WaveT(0);
    for (int64_t t0 = t_begin; t0 < t_end; t0 += TB) {
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
        [[maybe_unused]] OutT tail[TB];   // this lane's row of the tile
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
                    if constexpr (MODE != 1) {
                        const OutT y = (OutT)(fma(c4, z40, wv) * scale);
                        wb[s2 - 3] = y;
                        tail[s2 - 3] = y;
                    }
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
                    wb[s2 - 3] = (OutT)y4;
                    tail[s2 - 3] = (OutT)y4;
                }
                p1 = n1;
                p2 = n2;
                p3 = n3;
            }
        }
        if constexpr (MODE == 1) {                    // pass 1 of the time-split path keeps only the state
            wave_sync();
            continue;
        }
        wave_sync();
        store_window(t0, fits32 && full_rows && t0 > t_begin && t0 + TB <= t_end);
        wave_sync();
        // the tile's last d_lane samples in front of the row, for the next window (no other lane writes this row)
        if (lagged) {
#pragma unroll
            for (int j = 0; j < TB; ++j) *carry[j] = tail[j];
        }
        wave_sync();
    }
    if constexpr (MODE != 1) store_window(t_begin + (t_end - t_begin + TB - 1) / TB * TB, false);   // the lagging rest
    if constexpr (MODE == 1) {
        double* e = sp.states + ((size_t)unit * K + seg) * 8 * 64 + lane;
        e[0] = z10, e[64] = z11, e[128] = z20, e[192] = z21, e[256] = z30, e[320] = z31, e[384] = z40, e[448] = z41;
    }
}

// MODE 0: one unit per wave, unit = wave index. MODE 1 / 2: the two passes of the time-split path. MODE 3: the waves
// pull units from a queue (sp.order lists them longest first, sp.queue counts): a ragged batch whose waves would
// otherwise finish with the longest utterance of their SIMD.
template <typename WaveT, typename OutT, bool A2ZERO, int MODE>
__global__ __launch_bounds__((64 * waves_per_block<OutT, MODE>())) void k_erb_filterbank(const WaveT* __restrict__ wave,
                                                       const int64_t* __restrict__ offsets,
                                                       const double* __restrict__ coefs, int C,
                                                       int groups, int units, double* __restrict__ out,
                                                       float* __restrict__ alt, const int64_t* __restrict__ alt_off,
                                                       SplitArgs sp) {
    static_assert(MODE == 0 || A2ZERO, "the time-split and queue paths use the direct-form-II kernel");
    constexpr int WPB = waves_per_block<OutT, MODE>();
    __shared__ OutT tiles[WPB][64 * TileGeo<OutT>::PITCH_MAX];
    __shared__ double xss[WPB][TileGeo<OutT>::TB];
    __shared__ double msh[MODE == 2 ? WPB : 1][MODE == 2 ? 64 : 1][64];   // M of this wave's channels, [entry][lane]
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (tells the compiler it is wave-uniform)
    if constexpr (MODE == 3) {
        for (;;) {
            int t = 0;
            if (lane == 0) t = atomicAdd(sp.queue, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= units) return;
            filterbank_unit<WaveT, OutT, A2ZERO, MODE>(wave, offsets, coefs, C, groups, out, alt, alt_off, sp, sp.order[t], 0,
                                                       lane, tiles[wid], xss[wid], msh[0]);
        }
    } else {
        const int useg = blockIdx.x * WPB + wid;          // (utterance, group of 64 channels[, segment])
        const int K = MODE == 0 ? 1 : sp.K;
        const int unit = useg / K;
        if (unit >= units) return;                        // whole wave: the waves of a workgroup never meet at a barrier
        filterbank_unit<WaveT, OutT, A2ZERO, MODE>(wave, offsets, coefs, C, groups, out, alt, alt_off, sp, unit,
                                                   useg - unit * K, lane, tiles[wid], xss[wid], msh[MODE == 2 ? wid : 0]);
    }
}

template <typename WaveT, typename OutT>
void launch_fb(hipStream_t st, int units, bool a2zero, const void* wave, const int64_t* offsets, const double* coefs, int C,
               int groups, double* out, float* alt, const int64_t* alt_off, const SplitArgs& sp) {
    constexpr int WPB = waves_per_block<OutT>();
    const dim3 block(64 * WPB);
    if (sp.queue) {
        // ragged batch: persistent waves pulling units longest first. One wave per SIMD unless there are at least two
        // units per wave at two per SIMD: with fewer, the second wave of a SIMD only takes the balancing away
        // (5000 utterances of 1-4 s in launches of 2500: 24.9 ms with 2048 waves against 28.8; launches of 1000: 27.1
        // against 28.1 the other way round)
        const int waves = std::max(1, std::min(units, sp.qwaves > 0 ? sp.qwaves : (units >= 4096 ? 2048 : 1024)));
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, true, 3>), dim3((unsigned)((waves + WPB - 1) / WPB)), block, 0, st,
                           (const WaveT*)wave, offsets, coefs, C, groups, units, out, alt, alt_off, sp);
        return;
    }
    if (sp.K > 1) {
        // time-split path (a2zero only): pass 1 leaves the zero-state end states, pass 2 chains them and stores
        constexpr int WPB2 = waves_per_block<OutT, 2>();
        const size_t waves = (size_t)units * sp.K;
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, true, 1>), dim3((unsigned)((waves + WPB - 1) / WPB)), block, 0, st,
                           (const WaveT*)wave, offsets, coefs, C, groups, units, out, alt, alt_off, sp);
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, true, 2>), dim3((unsigned)((waves + WPB2 - 1) / WPB2)), dim3(64 * WPB2),
                           0, st, (const WaveT*)wave, offsets, coefs, C, groups, units, out, alt, alt_off, sp);
        return;
    }
    const dim3 grid((unsigned)((units + WPB - 1) / WPB));
    if (a2zero)
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, true, 0>), grid, block, 0, st, (const WaveT*)wave, offsets, coefs, C,
                           groups, units, out, alt, alt_off, sp);
    else
        hipLaunchKernelGGL((k_erb_filterbank<WaveT, OutT, false, 0>), grid, block, 0, st, (const WaveT*)wave, offsets, coefs,
                           C, groups, units, out, alt, alt_off, sp);
}

// M = T^L per channel for the time-split path. T is the zero-input transition of the eight state words
// (w1[-1], w1[-2], ..., w4[-1], w4[-2]) over one sample of the direct-form-II cascade the kernel runs:
//   w_k = in_k - a1 w_k[-1] - a2 w_k[-2],  out_k = w_k + c_k w_k[-1],  in_(k+1) = out_k,  in_1 = 0
void transition_power(const double* coef_row, int L, double* M) {
    typedef long double ld;
    const ld b0inv = 1.0L / (ld)coef_row[6];
    const ld a1 = (ld)coef_row[7] * b0inv, a2 = (ld)coef_row[8] * b0inv;
    const ld ck[4] = {(ld)coef_row[1] / (ld)coef_row[0], (ld)coef_row[2] / (ld)coef_row[0], (ld)coef_row[3] / (ld)coef_row[0],
                      (ld)coef_row[4] / (ld)coef_row[0]};
    ld T[8][8], R[8][8], X[8][8];
    for (int col = 0; col < 8; ++col) {
        ld v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nv[8];
        v[col] = 1;
        ld in = 0;
        for (int k = 0; k < 4; ++k) {
            const ld w = in - a1 * v[2 * k] - a2 * v[2 * k + 1];
            in = w + ck[k] * v[2 * k];
            nv[2 * k] = w;
            nv[2 * k + 1] = v[2 * k];
        }
        for (int r = 0; r < 8; ++r) T[r][col] = nv[r];
    }
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) R[r][c] = r == c ? 1 : 0;
    for (int bits = L; bits; bits >>= 1) {            // R = T^L by binary powering
        if (bits & 1) {
            for (int r = 0; r < 8; ++r)
                for (int c = 0; c < 8; ++c) {
                    ld acc = 0;
                    for (int q = 0; q < 8; ++q) acc += R[r][q] * T[q][c];
                    X[r][c] = acc;
                }
            memcpy(R, X, sizeof(R));
        }
        for (int r = 0; r < 8; ++r)
            for (int c = 0; c < 8; ++c) {
                ld acc = 0;
                for (int q = 0; q < 8; ++q) acc += T[r][q] * T[q][c];
                X[r][c] = acc;
            }
        memcpy(T, X, sizeof(T));
    }
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) M[r * 8 + c] = (double)R[r][c];
}

// Segments per utterance for this batch: 1 (plain kernel) unless the batch is too small to fill the chip.
int split_segments(int units, int64_t nmax, bool f32_out, int forced) {   // forced: -1 auto, 0 never, K >= 2 segments
    if (forced == 0 || nmax < 2 * SEG_ALIGN) return 1;
    // (measured, tools/k1_small_batch.py: with the float32 hand-off 192 units gain 1.7x, 384 units break even, 512 lose
    // 15 %; with float64 output, which is bound by its stores, 512 units lose 25 %)
    // (one or two utterances - `cnn eval` of a file, BASELINE's cfg1: 64 segments. The kernel pair costs ~2 L x 47 ns + 0.2 us
    // x K + 8 us, flat between K = 64 and 100 for 1 s: 56 -> 45 us; from 16 units on 32 segments are the better choice:
    // tools/k1_split_sweep.py, profiles/r05_k1_split_sweep.txt)
    const int kmax = units <= 4 ? 64 : 32;
    int K = forced >= 2 ? forced : (units <= (f32_out ? 320 : 256) ? std::min(kmax, 2048 / std::max(units, 1)) : 1);
    const int min_seg = units <= 4 ? 4 * SEG_ALIGN : 8 * SEG_ALIGN;                  // at least 128 / 256 samples per segment
    K = (int)std::min<int64_t>(K, (nmax + min_seg - 1) / min_seg);
    return std::max(K, 1);
}

}  // namespace

int f2_launch_filterbank(f2_ctx* ctx, const void* d_wave, int wave_dtype, const int64_t* d_offsets,
                         const int64_t* h_offsets, const double* d_coefs, int B, int C, double* d_gfb,
                         const f2_handoff* handoff, const int* d_uflag, const int* h_flag0) {
    const bool f32_out = handoff && handoff->f32;
    float* alt = f32_out ? handoff->d_x32 : nullptr;
    const int64_t* alt_off = f32_out ? handoff->d_x32_off : nullptr;
    const int groups = (C + 63) / 64;
    const int units = B * groups;
    // the coefficient rows of this call are mirrored on the host by f2_upload_coefs
    bool a2zero = ctx->coefs_host.size() == (size_t)C * 10;
    for (int c = 0; a2zero && c < C; ++c) a2zero = ctx->coefs_host[(size_t)c * 10 + 5] == 0.0;
    SplitArgs sp = {1, 0, nullptr, nullptr, nullptr, nullptr, d_uflag, ctx->opt_k1_qwaves};
    if (a2zero) {
        int64_t nmax = 0;
        for (int b = 0; b < B; ++b) nmax = std::max(nmax, h_offsets[b + 1] - h_offsets[b]);
        // Utterances the spectral kernel serves are skipped (flag 0) unless its accuracy guard hands them back, which is rare:
        // what this launch has to WORK on is known on the host - the utterances whose flag starts at 1 - and when those are a
        // handful of a large batch (the 1-2 % of a ragged corpus whose rows have too little padding for that route), a wave per
        // unit would walk a whole utterance serially while the chip idles (2.3 ms per 2500-utterance launch of the ragged
        // corpus for 1 % of its samples): the time-split path is chosen by the ACTIVE units then, with few enough segments
        // that the waves of skipped utterances (which return at once) stay cheap to launch.
        int active_units = units;
        if (d_uflag && h_flag0) {
            int na = 0;
            for (int b = 0; b < B; ++b) na += h_flag0[b] != 0;
            active_units = na * groups;
        }
        int K = split_segments(active_units > 0 ? active_units : units, nmax, f32_out, ctx->opt_k1_split);
        if (active_units != units && ctx->opt_k1_split < 2) K = (int)std::min<int64_t>(K, std::max<int64_t>(1, 65536 / std::max(units, 1)));
        if (K > 1) {
            sp.L = (int)(((nmax + K - 1) / K + SEG_ALIGN - 1) / SEG_ALIGN * SEG_ALIGN);
            sp.K = (int)((nmax + sp.L - 1) / sp.L);
        }
        if (sp.K > 1) {
            if (ctx->k1_mtab_L != sp.L || ctx->k1_mtab_coefs != ctx->coefs_host) {
                std::vector<double> m((size_t)C * 64);
                for (int c = 0; c < C; ++c) transition_power(&ctx->coefs_host[(size_t)c * 10], sp.L, &m[(size_t)c * 64]);
                F2_TRY(f2_reserve(ctx, ctx->k1_mtab, sizeof(double) * m.size()));
                F2_TRY(f2_upload_async(ctx, ctx->k1_mtab.ptr, m.data(), sizeof(double) * m.size()));
                ctx->k1_mtab_L = sp.L;
                ctx->k1_mtab_coefs = ctx->coefs_host;
            }
            F2_TRY(f2_reserve(ctx, ctx->k1_states, sizeof(double) * 8 * 64 * (size_t)units * sp.K));
            sp.states = (double*)ctx->k1_states.ptr;
            sp.mtab = (const double*)ctx->k1_mtab.ptr;
        } else {
            sp.K = 1;
            // lengths that differ a lot: hand the units out longest first instead of one fixed unit per wave
            int64_t total = 0;
            for (int b = 0; b < B; ++b) total += h_offsets[b + 1] - h_offsets[b];
            const bool want = ctx->opt_k1_queue >= 0 ? ctx->opt_k1_queue != 0 : (units > 1024 && (double)nmax * B > 1.2 * (double)total);
            if (want) {
                std::vector<int> order((size_t)units + 1);
                std::vector<int> by_len((size_t)B);
                for (int b = 0; b < B; ++b) by_len[(size_t)b] = b;
                std::stable_sort(by_len.begin(), by_len.end(), [&](int x, int y) {
                    return h_offsets[x + 1] - h_offsets[x] > h_offsets[y + 1] - h_offsets[y];
                });
                for (int i = 0; i < B; ++i)
                    for (int g = 0; g < groups; ++g) order[(size_t)i * groups + g] = by_len[(size_t)i] * groups + g;
                order[(size_t)units] = 0;     // the queue counter
                F2_TRY(f2_reserve(ctx, ctx->k1_order, sizeof(int) * order.size()));
                F2_TRY(f2_upload_async(ctx, ctx->k1_order.ptr, order.data(), sizeof(int) * order.size()));
                sp.order = (const int*)ctx->k1_order.ptr;
                sp.queue = (int*)ctx->k1_order.ptr + units;
            }
        }
    }
    F2_TRY(f2_prof_begin(ctx, F2_K_FILTERBANK));
    if (wave_dtype == F2_WAVE_I16 && !f32_out)
        launch_fb<int16_t, double>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb, nullptr, nullptr, sp);
    else if (wave_dtype == F2_WAVE_I16)
        launch_fb<int16_t, float>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb, alt, alt_off, sp);
    else if (!f32_out)
        launch_fb<double, double>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb, nullptr, nullptr, sp);
    else
        launch_fb<double, float>(ctx->stream, units, a2zero, d_wave, d_offsets, d_coefs, C, groups, d_gfb, alt, alt_off, sp);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(f2_prof_end(ctx, F2_K_FILTERBANK));
    return F2_OK;
}
