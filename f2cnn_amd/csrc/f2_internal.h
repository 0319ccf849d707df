// Internal declarations shared by the translation units of libf2cnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/f2cnn_hip.h"

struct f2_scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// tables of the spectral filterbank + envelope kernel (f2_spectral.hip) for one (coefficient table, length class)
struct f2_spec_tables {
    int log2h = 0, C = 0;
    std::vector<double> coefs;
    int64_t tpitch = 0;
    f2_scratch hu, e, lgroup, e64;
};
#define F2_SPECTRAL_MIN_LOG2H 12   // rows of 4097 ... 65536 samples
#define F2_SPECTRAL_MAX_LOG2H 15

struct f2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    int num_cus = 0;
    char err[512] = {0};
    // grow-only device scratch areas (never freed before ctx destroy; stream-ordered reuse only)
    f2_scratch coefs;      // filter coefficients of the current call
    f2_scratch offsets;    // ragged offsets of the current call
    f2_scratch stage_in;   // F2_MEM_HOST staging
    f2_scratch stage_out;
    f2_scratch stage_aux;
    f2_scratch work;       // intermediates (GFB between K1 and K2, activations, ...)
    f2_scratch work2;
    f2_scratch xbuf;       // window tensor chunk between K3 and K4
    f2_scratch dense_in;   // conv4 outputs (+ dense1 outputs) of the windows of several utterances: one dense launch for all
    f2_scratch gather_log; // ln of the envelope samples a chunk of every-sample windows touches + column min / max (f2_gather.hip)
    f2_scratch tw[2][16];  // FFT twiddle tables, [precision][log2 H], built on first use
    f2_scratch tw_fl[16];         // twiddle tables of f2_envelope_flagged.hip, by log2 H
    f2_scratch tw_p3[2];   // the same for the three-pass plan of H = 8192 (f2_envelope_p3.hip), [precision]
    f2_scratch tw_large[2][24];   // same for the global-memory transform of long rows
    f2_scratch tw_split[24];      // tables of the four-step transform (f2_envelope_split.hip), by log2 H
    std::vector<int> pair_list_host[2];   // what pair_list currently holds (skip the upload when equal)
    f2_scratch tw_pair[2], pair_list[2];   // two-sub-row transform of 16385..65536-sample rows (f2_envelope_pair.hip): tables, utterance lists
    f2_scratch work3;             // utterance lists of the four-step launches
    f2_scratch handoff, handoff_off;   // float32 hand-off of long rows (f2_plan_handoff)
    std::vector<int64_t> handoff_off_host;
    f2_scratch k1_states, k1_mtab;     // time-split filterbank (small batches): segment end states, T^L per channel
    f2_scratch k1_order;               // ragged batches: unit order (longest first) + the queue counter
    int k1_mtab_L = 0;                 // segment length k1_mtab was built for ...
    std::vector<double> k1_mtab_coefs; // ... and the coefficient rows
    std::vector<int64_t> offsets_host;  // what ctx->offsets currently holds (skip re-upload when equal)
    std::vector<double> coefs_host;     // what ctx->coefs currently holds
    bool prof_on = false;
    struct prof_span {
        hipEvent_t first, second;   // start, stop
        bool closed;                // stop has been recorded (false when the launch in between failed)
    };
    std::vector<prof_span> prof[F2_K_COUNT];                         // one span per launch group
    std::vector<hipEvent_t> prof_pool;                               // recycled events
    // spectral path (f2_spectral.hip)
    std::vector<f2_spec_tables> spec_tabs;
    f2_scratch spec_x, spec_rho;          // utterance spectra, per-row digits of the launch in flight
    f2_scratch spec_xpart;                // float64 partial spectra of decimated (long) utterances
    f2_scratch spec_meta, spec_uflag;     // [initial flags (B) | utterance lists]; the flags the kernels update
    f2_scratch spec_gdump;                // [B][C][4] floats, option "spectral_guard_dump"
    size_t spec_gdump_rows = 0;           // B x C of the call that filled it
    std::vector<int> spec_meta_host;      // what spec_meta currently holds
    int spec_coefs_ok = -1;               // coefs_host eligible for the spectral kernel: -1 not decided, 0, 1
    int spec_min_pad = 64;                // ... and the padding samples its accuracy guard needs (ringing peak of the slowest channel)
    size_t spec_last_B = 0;               // batch size of the last fused call that used the spectral kernel (0: none)
    f2_scratch tw_sp[2][16];              // its twiddle tables, [precision][log2 H]
    f2_scratch spec_lptab;                // low-pass powers per thread (lowpass_pairs_store_tab) ...
    double spec_lptab_a1 = 0.0;           // ... for this a1
    int spec_lptab_nt = 0;                // ... and workgroup size
    // options (f2_ctx_set_option); -1 = decide from the batch
    int opt_spectral = 1;                 // route eligible utterances of the fused call through the spectral kernel
    int opt_spectral_min_rows = 4096;     // ... when the call has at least this many eligible rows (utterances x channels)
    float opt_spectral_tol = 4e-6f;       // accuracy guard: padding residual / maximum of the delivered row that flags an utterance
    int opt_spectral_min_pad = -1;        // padding samples a row needs for that route: -1 = from the coefficient table, else >= 64
    int opt_spectral_guard_dump = 0;      // diagnostic: keep the guard's per-row values of the last fused call (f2_spectral_guard_read)
    int opt_k1_split = -1;                // segments of the time-split filterbank (0 = never, >= 2 = force)
    int opt_k1_queue = -1;                // unit queue of the filterbank for ragged batches (0 / 1)
    int opt_k1_qwaves = 0;                // waves of the queue launch (0 = from the batch)
    int opt_env_pair = 1;                 // on-chip kernel for rows of 32769..65536 samples
    int opt_cnn_bf16x3 = 1;               // conv2 .. conv4 + dense1 on the fp16 matrix cores, operands split in two pieces (3 MFMAs per product; "cnn_f16x3")
    int opt_cnn_ws = 1;                   // ... with the weights of each wave's role held in registers (f2_cnn_ws.hip; windows of 10 / 11 rows)
    int opt_cnn_ws_dense = 1;             // ... and dense1 with 96 windows per weight fragment, loads waited for by hand (k_dense1_ws)
    int opt_gather_blocked = 1;           // every-sample windows: logarithm once per sample, blocks of 32 windows (0: one workgroup per window)
    int opt_env_plan4 = 0;                // four-pass plan for every 1 s row (default: three passes where measured faster)
    // pinned staging of small host -> device uploads (f2_upload_async): a ring of page-locked memory the copies read from,
    // so that a call with device pointers only enqueues (the caller's / a local's array is copied on the host at once)
    char* up_ring = nullptr;
    size_t up_cap = 0, up_head = 0;
    struct up_span {
        size_t begin, end;
        hipEvent_t done;
    };
    std::vector<up_span> up_inflight;                  // oldest first
    struct up_side {
        void* ptr;
        size_t cap;
        hipEvent_t done;   // recorded behind the last copy that read from the buffer
        bool busy;         // `done` not yet seen reached
    };
    std::vector<up_side> up_big;   // grow-only page-locked side buffers of uploads too large for the ring, reused once copied
    f2_scratch flags;      // small device words (error flags)
    int* host_flags = nullptr;  // pinned mirror
};

#include "f2_cnn_split.h"
// offsets (floats) into f2_cnn::sbias: conv2's biases x sa_3; conv3's x sa_3 sb_3 (accumulator-initial form) and x sa_4 (epilogue
// form); conv4's x sa_dense1
enum { F2_SB_B2 = 0, F2_SB_B3I = 64, F2_SB_B3F = 128, F2_SB_B4 = 192, F2_SB_FLOATS = 256 };

struct f2_cnn {
    int rows = 0, channels = 0, flat = 0;
    int dev = 0;
    float* blob = nullptr;       // all tensors, device
    size_t off[12] = {0};        // element offsets of the 12 tensors in `blob`
    const float* t(int i) const { return blob + off[i]; }
    uint16_t* blob16 = nullptr;  // conv2 .. conv4 and dense1 kernels, scaled and split into two fp16 pieces (f2_cnn_split.h, k_*_h16x3)
    f2_split_scales sc = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};   // the power-of-two scales that go with them
    float c2_true = 1.f, sa_d1 = 1.f;   // 1 / (sa_2 sb_2): conv2's epilogue multiplier for outputs in true units; dense1's input scale
    float* sbias = nullptr;      // biases in the scaled units the split kernels' epilogues use (F2_SB_* offsets)
    size_t off16[4] = {0};       // element offsets of the four layers in `blob16`
    const void* zeros = nullptr; // 256 zero bytes behind them (source of the padding pixels of f2_cnn_ws.hip's LDS-DMA loads)
    // f2_cnn_create's self-check of the weight-stationary kernels (hand-placed s_waitcnt around inline-asm loads: correct only
    // while the register allocator of the hipcc that built the library leaves those registers alone) against the per-tile
    // split-bf16 kernels on a fixed batch; a kernel that disagrees is not used with this network
    bool ws_ok = true, ws_dense_ok = true;
    float ws_check_diff = -1.f, ws_dense_check_diff = -1.f;   // max |score difference| measured (-1: not applicable)
};

// activation workspace (floats) the CNN needs per window
size_t f2_cnn_workspace_floats(const f2_cnn* cnn);

extern char g_f2_err[512];

int f2_fail(f2_ctx* ctx, int code, const char* fmt, ...);
int f2_reserve(f2_ctx* ctx, f2_scratch& s, size_t bytes);
// host -> device copy on the context's stream that does not wait for it: the bytes are copied into page-locked staging
// memory first (ring for small arrays, a one-off buffer for large tables), `src` may be reused at once
int f2_upload_async(f2_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int f2_upload_offsets(f2_ctx* ctx, const int64_t* offsets, int B);
int f2_upload_coefs(f2_ctx* ctx, const double* coefs, int C);

#define F2_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return f2_fail((ctx), F2_ERR_HIP, "%s -> %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                \
    } while (0)

#define F2_CHECK(ctx, cond, code, ...)                        \
    do {                                                      \
        if (!(cond)) return f2_fail((ctx), (code), __VA_ARGS__); \
    } while (0)

#define F2_TRY(expr)              \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != F2_OK) return rc_; \
    } while (0)

// Profiling bracket: f2_prof_begin before the launch(es) of one kernel id, f2_prof_end after. A span whose launch
// failed in between is never closed: it is recycled by the next f2_prof_begin / f2_prof_reset and skipped by f2_prof_get.
int f2_prof_begin(f2_ctx* ctx, int kernel_id);
int f2_prof_end(f2_ctx* ctx, int kernel_id);

// kernel arguments shared by the envelope kernels
struct f2_env_params {
    const double* gfb;
    double* env;
    const int64_t* offsets;
    const int* ulist;  // utterances served by this launch (NULL: identity)
    const int* uflag;  // per utterance of the batch, or NULL: rows of utterances whose flag is 0 are left alone
    int C;
    int lpf;
    int f32_in;        // input rows are float32 at the start of their float64 slot (hand-off from K1)
    unsigned long long* stamps;   // diagnostic build only
    double b0, a1;     // y[n] = b0 (e[n] + e[n-1]) - a1 y[n-1]
};
// H = 8192 rows (the 1 s / 16 kHz row) with the three-pass plan 16-32-16: f2_envelope_p3.hip compiles f2_envelope.hip
// a second time with F2_PLAN13_PASSES = 3 - in its own namespace, the radix plans being compile-time functions of the
// macro - and exports only this launcher. Measured against the four-pass plan: 8 % faster without the low-pass, 6 % with
// the float64 transform, 4 % slower with the float low-pass; f2_launch_envelope picks per call.
int f2_launch_envelope13_p3(f2_ctx* ctx, const f2_env_params& P, int precision, unsigned rows);
// f2_envelope_flagged.hip: the utterances P.ulist[0..nutt) of length class log2h whose P.uflag entry is set (float FFT)
int f2_launch_envelope_flagged(f2_ctx* ctx, const f2_env_params& P, int log2h, unsigned nutt);

// longest row (2^22 samples = 262 s at 16 kHz) the global-memory envelope path accepts
// rows between the LDS limit and 262144 samples: four-step transform with LDS-resident 4096-point parts
bool f2_envelope_split_supports(int log2h, int precision);
int f2_launch_envelope_split(f2_ctx* ctx, const double* d_gfb, double* d_env, const int64_t* d_offsets, const int* utts,
                             int nutt, int log2h, int C, int lpf, double b0, double a1, const float* d_x32,
                             const int64_t* d_x32_off, const int* d_uflag = nullptr);
// rows of 32769..65536 samples, float transforms, input not aliased with the output: two LDS-resident sub-rows per workgroup
bool f2_envelope_pair_supports(int log2h, int precision);
int f2_launch_envelope_pair(f2_ctx* ctx, const double* d_gfb, double* d_env, const int64_t* d_offsets,
                            const int64_t* h_offsets, const int* utts, int nutt, int log2h, int C, int lpf, double b0,
                            double a1, const float* d_x32, const int64_t* d_x32_off, const int64_t* h_x32_off,
                            const int* d_uflag = nullptr);
#define F2_MAX_LOG2M_LARGE 22
int f2_launch_envelope_large(f2_ctx* ctx, const double* d_x, double* d_y, int64_t n, int C, int lpf, double b0,
                             double a1, int precision);

static inline int f2_log2_ceil(int64_t n) {
    int k = 0;
    while ((int64_t(1) << k) < n) ++k;
    return k;
}

// ---- launchers implemented in the kernel translation units (device pointers only) ----
// How the filterbank hands its rows to the envelope kernels inside one call (decided from the utterance lengths):
// float64 rows in the output buffer, or float32 - at the start of each row's float64 slot for rows the LDS-resident
// kernel takes (<= 32768 samples), compact (C, n) float rows in a scratch buffer for the four-step path (whose last
// pass writes float64 results over the slot while other workgroups still read their inputs).
struct f2_handoff {
    bool f32 = false;
    float* d_x32 = nullptr;                 // scratch of the long rows, or NULL when there are none
    const int64_t* d_x32_off = nullptr;     // device, per utterance: float offset into d_x32, -1 = in the row's own slot
    const int64_t* h_x32_off = nullptr;     // the same on the host (owned by the context, valid until the next plan)
};
int f2_plan_handoff(f2_ctx* ctx, const int64_t* h_offsets, int B, int C, int precision, bool want_gfb, f2_handoff* plan);
// d_uflag (device, B ints) != NULL: only utterances whose flag is non-zero are processed (the rest were served by the
// spectral kernel); the flags may be written by earlier launches on the stream.
// h_flag0 (host, B ints, with d_uflag): the flags as they are before the spectral kernel runs - the utterances this launch
// certainly has to process (its launch shape is chosen for them; any utterance may still be handed back by the guard).
int f2_launch_filterbank(f2_ctx* ctx, const void* d_wave, int wave_dtype, const int64_t* d_offsets,
                         const int64_t* h_offsets, const double* d_coefs, int B, int C, double* d_gfb,
                         const f2_handoff* handoff = nullptr, const int* d_uflag = nullptr, const int* h_flag0 = nullptr);
int f2_launch_envelope(f2_ctx* ctx, const double* d_gfb, const int64_t* d_offsets, const int64_t* h_offsets,
                       int B, int C, int lpf, double cutoff_hz, int precision, double* d_env,
                       const f2_handoff* handoff = nullptr, const int* d_uflag = nullptr, const int* h_flag0 = nullptr);
// Spectral filterbank + envelope (f2_spectral.hip): which utterances / coefficient tables it serves, and the launch for
// the utterances d_ulist[0..nutt) (all of length class log2h). Rows that fail its accuracy guard set d_uflag[b].
bool f2_spectral_supports_len(int64_t n, int min_pad);
bool f2_spectral_supports_coefs(const std::vector<double>& coefs, int C, std::vector<int>* Lgroup, int* min_pad);
int f2_launch_spectral(f2_ctx* ctx, const void* d_wave, int wave_dtype, const int64_t* d_offsets, const double* d_coefs,
                       int C, const int* d_ulist, int nutt, int64_t min_n /* shortest row of the group */, int log2h, int lpf,
                       double cutoff_hz, double* d_env, int* d_uflag);
// d_centers == NULL: window e is centred at first_center + e
int f2_launch_gather(f2_ctx* ctx, const double* d_env, int C, int64_t N, const int64_t* d_centers,
                     int64_t first_center, int64_t n_windows, int radius, int step, int normalize, float* d_out, int* d_flag);
// weight-stationary split-bf16 convolutions (f2_cnn_ws.hip): windows whose pooled conv2 output has four rows
bool f2_cnn_ws_supported(int rows, int channels);
int f2_launch_dense1_ws(f2_ctx* ctx, const f2_cnn* cnn, const float* a4, int64_t n, int K, float* a5);
int f2_launch_cnn_ws(f2_ctx* ctx, const f2_cnn* cnn, const float* d_x, int64_t n, void* a2s, float* a4);
// runs the network on n windows (n <= chunk the workspace was sized for); d_ws: n * workspace floats
int f2_launch_cnn(f2_ctx* ctx, const f2_cnn* cnn, const float* d_x, int64_t n, float* d_ws, float* d_scores,
                  uint8_t* d_labels);
// the two halves of f2_launch_cnn (f2_cnn.hip): convolutions per chunk of windows, dense layers over several chunks at once
int f2_launch_cnn_convs(f2_ctx* ctx, const f2_cnn* cnn, const float* d_x, int64_t n, float* d_ws, float* a4);
int f2_launch_cnn_dense(f2_ctx* ctx, const f2_cnn* cnn, const float* a4, int64_t n, float* a5, float* d_scores, uint8_t* d_labels);
size_t f2_cnn_flat_floats(const f2_cnn* cnn);    // floats per window of the conv4 output
size_t f2_cnn_dense_floats(const f2_cnn* cnn);   // ... plus dense1's output
