/*
 * f2cnn_hip.h -- C ABI of libf2cnn_hip.so: the MI355X (gfx950) implementation of the F2CNN hot path.
 *
 * The reference (tictacmenthe/F2CNN) is pure Python and has no FFI; its boundary for this path is a
 * set of NumPy-in / NumPy-out functions. Each entry point below replaces the body of one of them
 * (file:line relative to the reference tree) and is what a ctypes binding in the reference would
 * call (see INTEGRATION.md for the stub). Plain pointers and sizes only; no torch / HIP types.
 *
 * Conventions
 *   - every function returns F2_OK (0) or a negative f2_status; f2_last_error() gives the text.
 *   - `mem_space` says whether DATA pointers (wave, gfb, env, x, scores, ...) are host or device
 *     pointers. Small metadata arrays (offsets, coefs, centers) are ALWAYS host pointers.
 *   - F2_MEM_HOST calls stage through device memory and return when the result is in the host
 *     buffer. F2_MEM_DEVICE calls enqueue on the context's stream and return immediately
 *     (f2_ctx_synchronize() or a stream-ordered consumer to wait): the small per-batch arrays they
 *     upload (offsets, utterance lists, ...) are staged through page-locked memory owned by the
 *     context, so a new batch shape does not wait for the stream either. What does wait: a scratch
 *     buffer that has to grow (the first call of a size), and the calls that hand an error flag of
 *     the device back (f2_gather_windows with normalisation, f2_eval_*: F2_ERR_NONPOSITIVE).
 *   - ragged batches: utterance b has n_b = offsets[b+1]-offsets[b] samples; its wave starts at
 *     wave + offsets[b]; its (C, n_b) C-order float64 matrix starts at out + C*offsets[b]. For a
 *     uniform batch this is the plain [B][C][N] layout, and each utterance's block is bit-for-bit the
 *     payload of the reference's .GFB.npy / .ENV1.npy file (row 0 = highest centre frequency).
 *   - one host thread per context; a context owns one device, one stream and its scratch memory.
 */
#ifndef F2CNN_HIP_H
#define F2CNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct f2_ctx f2_ctx;
typedef struct f2_cnn f2_cnn;

typedef enum {
    F2_OK = 0,
    F2_ERR_INVALID = -1,      /* bad argument (null pointer, negative size, ...)              */
    F2_ERR_HIP = -2,          /* a HIP runtime call failed (no device, launch failure, ...)    */
    F2_ERR_UNSUPPORTED = -3,  /* valid request this build cannot serve (e.g. row too long)     */
    F2_ERR_NOMEM = -4,        /* device or host allocation failed                              */
    F2_ERR_NONPOSITIVE = -5   /* normalizeInput met a value <= 0 (reference: ValueError)       */
} f2_status;

/* F2_MEM_HOST_ASYNC: host pointers like F2_MEM_HOST, but the call returns as soon as the copies and kernels are
 * queued on the context's stream; the host buffers must stay valid (and should be page-locked: f2_host_alloc, so
 * that the copies really are asynchronous) until f2_ctx_synchronize() returns. The file drivers use it with two
 * contexts so that batch k's device-to-host copy runs beside batch k+1's host-to-device copy and kernels.
 * Accepted by f2_erb_filterbank_batch, f2_envelope_batch and f2_filterbank_envelope_fused. */
enum { F2_MEM_HOST = 0, F2_MEM_DEVICE = 1, F2_MEM_HOST_ASYNC = 2 };
/* Alignment: device buffers need the natural alignment of their element type only (hipMalloc gives far more).
 * Utterance lengths are arbitrary: rows of the (C, n) matrices start wherever C-order puts them, and the kernels keep
 * their stores on whole 128-byte lines and their loads wide for any n (no padding of n, no pitch parameter). */
enum { F2_WAVE_I16 = 0, F2_WAVE_F64 = 1 };
/* arithmetic of the Hilbert FFT: F2_FFT_F32 (default; 3.6e-7 max-norm error) or
 * F2_FFT_F64 (reference-grade, slower). The IIR recurrences are float64 in both. */
enum { F2_FFT_F32 = 0, F2_FFT_F64 = 1 };

/* ---- library / context -------------------------------------------------------------------- */
int f2_version(void);   /* 100 * major + minor; 101 added f2_eval_batch, 102 f2_host_alloc + F2_MEM_HOST_ASYNC, 103 f2_ctx_set_option, 105 f2_spectral_guard_read + f2_cnn_get_info */
int f2_device_count(int* count);
int f2_ctx_create(int device, f2_ctx** ctx);
int f2_ctx_destroy(f2_ctx* ctx);
int f2_ctx_synchronize(f2_ctx* ctx);
/* adopt an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int f2_ctx_set_stream(f2_ctx* ctx, void* hip_stream);
void* f2_ctx_get_stream(f2_ctx* ctx);
/* text of the last error on this context (ctx == NULL: last context-less error) */
const char* f2_last_error(f2_ctx* ctx);
/* Per-context tuning switches (the reference has none: its only knobs are the CLI arguments). Unknown key:
 * F2_ERR_INVALID. Keys (value -1 = decide from the batch, where stated):
 *   "spectral"       1 (default) / 0   f2_filterbank_envelope_fused serves eligible utterances with the one-kernel
 *                                      spectral path; 0 = always filterbank kernel + envelope kernel
 *   "spectral_min_rows"  rows (utterances x channels) a call needs before that path is used (default 4096: below,
 *                        the serial filter-state kernel is not hidden and the time-split filterbank kernel is faster)
 *   "spectral_tol"   accuracy guard of that path (default 4e-6): padding-region residual, relative to the maximum of the row as
 *                    delivered (the low-passed row when lpf != 0), that sends an utterance back
 *   "spectral_min_pad"  zero-padding samples (2^k - n) a row needs for that path: -1 (default) = what the slowest channel's
 *                    ringing needs to reach its peak, from the coefficient table (256 for the reference's 100 Hz .. 8 kHz
 *                    bank), so that the guard sees the error it has to judge; an explicit value >= 64 for experiments
 *   "spectral_guard_dump"  1 / 0 (default)   keep the guard's per-row values (f2_spectral_guard_read)
 *   "k1_split"       -1 / 0 / K >= 2   time-split filterbank for small batches: auto / never / K segments
 *   "k1_queue"       -1 / 0 / 1        unit queue of the filterbank for ragged batches
 *   "k1_qwaves"      0 / n             waves of the queue launch (0 = from the batch)
 *   "env_pair"       1 / 0             on-chip envelope kernel for rows of 32769..65536 samples
 *   "env_plan4"      0 / 1             four-pass transform plan for every 8193..16384-sample row
 *   "cnn_bf16x3"     1 (default) / 0   conv2..conv4 of f2_cnn_* / f2_eval_* on the bf16 matrix cores with both operands
 *                                      split in two bf16 pieces (three MFMAs per product, float32 accumulation: scores within
 *                                      1e-6 of the float32 matrix path, 2.2 x its speed); 0 = v_mfma_f32_32x32x2_f32 throughout
 *   "cnn_ws"         1 (default) / 0   with "cnn_bf16x3", windows of 10 / 11 rows (the reference's 11 x C): persistent weight-
 *                                      stationary kernels (each wave keeps the weights of its role in registers, conv1 on the
 *                                      matrix cores too, one barrier per tile); 0 = one workgroup per tile, weights re-read
 *   "cnn_ws_dense"   1 (default) / 0   with "cnn_ws": dense1 on 96-window tiles (a weight fragment feeds nine MFMAs), its loads
 *                                      issued and waited for by hand; 0 = the 64-window kernel of "cnn_bf16x3"
 *   "gather_blocked" 1 (default) / 0   every-sample normalised windows (f2_gather_windows without centres, f2_eval_*): logarithm
 *                                      once per envelope sample and blocks of 32 consecutive windows, bit-identical to 0 = one
 *                                      workgroup per window
 * Read-only (f2_ctx_get_option): "spectral_routed" = utterances of the last fused call that went through the spectral
 * kernel, "spectral_flagged" = those of them its accuracy guard handed back to the two-kernel route (waits for the
 * stream); "spectral_routed_samples" / "spectral_flagged_samples" = the same in samples.
 * Two contexts on two host threads choose independently. */
int f2_ctx_set_option(f2_ctx* ctx, const char* key, double value);
int f2_ctx_get_option(f2_ctx* ctx, const char* key, double* value);
/* Diagnostic of the spectral path's accuracy guard (no reference counterpart; used by tests/diag/guard_search.py). With
 * option "spectral_guard_dump" = 1 the last f2_filterbank_envelope_fused call that took the spectral route keeps, per
 * (utterance b, channel c) row in batch order, four floats {maximum of |analytic signal| inside the row, maximum of the
 * padding-region residual, maximum of the low-passed row (0 without low-pass), 1 if this row tripped the guard}; rows the
 * spectral kernel did not serve read as NaN. Copies min(rows, available) rows to `out` (host) and waits for the stream. */
int f2_spectral_guard_read(f2_ctx* ctx, float* out, int64_t rows, int64_t* rows_available);

/* ---- device memory + timing helpers (so a host language needs no HIP binding of its own) ---- */
int f2_dev_malloc(f2_ctx* ctx, size_t bytes, void** dptr);
int f2_dev_free(f2_ctx* ctx, void* dptr);
int f2_dev_memset(f2_ctx* ctx, void* dptr, int value, size_t bytes);
/* page-locked host memory for the staging buffers of F2_MEM_HOST / F2_MEM_HOST_ASYNC calls (the reference's
 * numpy.save / numpy.load buffers, GammatoneFiltering.py:61-62, EnvelopeExtraction.py:80,94-95) */
int f2_host_alloc(f2_ctx* ctx, size_t bytes, void** hptr);
int f2_host_free(f2_ctx* ctx, void* hptr);
int f2_memcpy_h2d(f2_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int f2_memcpy_d2h(f2_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int f2_event_create(f2_ctx* ctx, void** event);
int f2_event_destroy(f2_ctx* ctx, void* event);
int f2_event_record(f2_ctx* ctx, void* event);                 /* on the context's stream */
int f2_event_elapsed_ms(f2_ctx* ctx, void* start, void* stop, float* ms); /* waits for `stop` */
int f2_event_query(f2_ctx* ctx, void* event, int* done);       /* does not wait: *done = 1 once the stream has passed it */

/* ---- per-kernel timing (HIP events recorded around every kernel launch on the context's stream) ----
 * Kernel ids: F2_K_* below. f2_prof_get waits for the stream, returns the number of launches of that
 * kernel since the last f2_prof_enable(ctx, 1) / f2_prof_reset and their summed device time. */
enum {
    F2_K_FILTERBANK = 0,   /* k_erb_filterbank                                                       */
    F2_K_ENVELOPE = 1,     /* k_envelope and its long-row variants                                   */
    F2_K_GATHER = 2,
    F2_K_CNN = 3,
    F2_K_FUSED = 4,        /* k_spectral_envelope: filterbank + envelope of a row in ONE kernel      */
    F2_K_SPECTRUM = 5,     /* k_utterance_spectrum: float64 transform of each utterance, once        */
    F2_K_TAIL = 6,         /* k_tail_state: filter state at the end of each row                      */
    F2_K_COUNT = 7
};
int f2_prof_enable(f2_ctx* ctx, int on);
int f2_prof_reset(f2_ctx* ctx);
int f2_prof_get(f2_ctx* ctx, int kernel_id, int* launches, float* total_ms);
const char* f2_prof_kernel_name(int kernel_id);

/* ---- K1: ERB gammatone filterbank ------------------------------------------------------------
 * Replaces gammatone/filters.py:195-239 erb_filterbank (called from
 * scripts/processing/GammatoneFiltering.py:42-47 GetFilteredOutputFromArray and
 * scripts/CNN/Evaluating.py:52), batched over utterances.
 *   wave     int16 (F2_WAVE_I16) or float64 (F2_WAVE_F64) samples, ragged by `offsets`
 *   offsets  host, B+1 entries, offsets[0] == 0, non-decreasing
 *   coefs    host, (C,10) float64 rows [A0,A11,A12,A13,A14,A2,B0,B1,B2,gain] (make_erb_filters)
 *   gfb      out, float64, C*offsets[B] elements
 */
int f2_erb_filterbank_batch(f2_ctx* ctx, const void* wave, int wave_dtype, const int64_t* offsets,
                            const double* coefs, int B, int C, double* gfb, int mem_space);

/* ---- K2: Hilbert-magnitude envelope + optional 1st-order Butterworth low-pass ----------------
 * Replaces scripts/processing/EnvelopeExtraction.py:51-67 ExtractEnvelopeFromMatrix (with
 * paddedHilbert :20-36 and lowPassFilter :39-48), batched. lpf == 0: magnitude only; otherwise
 * butter(1, cutoff_hz/8000) (8000 hard-coded as in the reference) applied from zero state.
 *   gfb / env  float64, ragged (C, n_b) blocks as above (env may alias gfb)
 */
int f2_envelope_batch(f2_ctx* ctx, const double* gfb, const int64_t* offsets, int B, int C, int lpf,
                      double cutoff_hz, int fft_precision, double* env, int mem_space);

/* ---- K1+K2 without the float64 GFB round trip through HBM -------------------------------------
 * `prepare filter` + `prepare envelope` in one call (GammatoneFiltering.py:69-78 followed by
 * EnvelopeExtraction.py:101-117). gfb_or_null != NULL additionally emits the filterbank output
 * (needed when .GFB.npy files must be written). */
int f2_filterbank_envelope_fused(f2_ctx* ctx, const void* wave, int wave_dtype, const int64_t* offsets,
                                 const double* coefs, int B, int C, int lpf, double cutoff_hz,
                                 int fft_precision, double* env, double* gfb_or_null, int mem_space);

/* ---- K3: window gather (+ per-window log min-max normalisation) ------------------------------
 * Replaces the Python gathers of scripts/processing/InputGenerator.py:73-80 (centers given,
 * normalize = 0, output cast to float32 as at :83) and scripts/CNN/Evaluating.py:76-80
 * (centers == NULL: centre_i = radius*step + i for i < n_windows; normalize = 1 applies
 * scripts/CNN/Training.py:13-28 normalizeInput in float64 before the float32 cast).
 *   env      (C, N) float64 C-order
 *   out      (n_windows, 2*radius+1, C) float32
 * Returns F2_ERR_NONPOSITIVE if normalize != 0 and a window holds a value <= 0 (reference raises
 * ValueError), F2_ERR_INVALID if a window reaches outside [0, N).
 */
int f2_gather_windows(f2_ctx* ctx, const double* env, int C, int64_t N, const int64_t* centers,
                      int64_t n_windows, int radius, int step, int normalize, float* out, int mem_space);

/* ---- K4: CNN forward ---------------------------------------------------------------------------
 * Replaces keras model.predict + the label rule of scripts/CNN/Evaluating.py:85-87 for the
 * architecture built at scripts/CNN/Training.py:93-114.
 * f2_cnn_create copies 12 host tensors in Keras layouts, in this order:
 *   conv1 kernel (3,3,1,32) bias (32) | conv2 (3,3,32,32),(32) | conv3 (3,3,32,64),(64) |
 *   conv4 (3,3,64,64),(64) | dense1 (F,516),(516) | dense2 (516,2),(2),  F = flatten size for (rows, channels)
 * f2_cnn_forward: x (n, rows, channels) float32 -> scores (n,2) softmax float32 and
 * labels[i] = scores[i][1] > scores[i][0] (ties -> 0). scores or labels may be NULL.
 */
int f2_cnn_create(f2_ctx* ctx, const float* const* tensors, int rows, int channels, f2_cnn** cnn);
int f2_cnn_destroy(f2_ctx* ctx, f2_cnn* cnn);
/* What f2_cnn_create found out about this network (no reference counterpart). Keys: "flat" (flatten size); "ws_ok" /
 * "ws_dense_ok" = 1 if the weight-stationary convolution / dense1 kernels (option "cnn_ws" / "cnn_ws_dense") serve this
 * network: its shape qualifies AND they reproduced the per-tile kernels' scores on f2_cnn_create's self-check batch (their
 * hand-placed memory waits are only valid for the register allocation of the compiler they were validated with; a library
 * built by another hipcc that fails the check falls back to the per-tile kernels and says so on stderr);
 * "ws_check_diff" / "ws_dense_check_diff" = the score differences measured (-1: not applicable). */
int f2_cnn_get_info(f2_ctx* ctx, const f2_cnn* cnn, const char* key, double* value);
int f2_cnn_forward(f2_ctx* ctx, const f2_cnn* cnn, const float* x, int64_t n, float* scores,
                   uint8_t* labels, int mem_space);

/* ---- `cnn eval` device pipeline -----------------------------------------------------------------
 * scripts/CNN/Evaluating.py:42-87 for one utterance with every intermediate kept in HBM:
 * filterbank -> envelope -> every-sample window gather + normalise -> CNN -> labels.
 * n_windows_out receives N - (2*radius+1)*step (Evaluating.py:73). env_or_null (C,N) float64,
 * scores_or_null (n,2), labels_or_null (n) are optional outputs in `mem_space`.
 */
int f2_eval_utterance(f2_ctx* ctx, const f2_cnn* cnn, const void* wave, int wave_dtype, int64_t N,
                      const double* coefs, int C, int lpf, double cutoff_hz, int fft_precision,
                      int radius, int step, double* env_or_null, float* scores_or_null,
                      uint8_t* labels_or_null, int64_t* n_windows_out, int mem_space);

/* The same pipeline for a ragged batch of utterances (scripts/CNN/Evaluating.py:138-177 EvaluateRandom evaluates a
 * list of files with one model): filterbank and envelope run once for the whole batch - a single utterance only
 * fills two wavefronts of the filterbank kernel - then windows + CNN utterance by utterance. Utterance b has
 * nb_b = max(0, n_b - (2*radius+1)*step) windows; scores / labels are the concatenation over b in batch order
 * (sum nb_b rows), both optional, in `mem_space`.
 */
int f2_eval_batch(f2_ctx* ctx, const f2_cnn* cnn, const void* wave, int wave_dtype, const int64_t* offsets,
                  const double* coefs, int B, int C, int lpf, double cutoff_hz, int fft_precision, int radius,
                  int step, float* scores_or_null, uint8_t* labels_or_null, int mem_space);

#ifdef __cplusplus
}
#endif
#endif /* F2CNN_HIP_H */
