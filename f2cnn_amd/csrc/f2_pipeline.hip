// C-ABI entry points for the window gather (K3), the CNN forward (K4) and the `cnn eval` device pipeline
// (scripts/CNN/Evaluating.py:42-87): host/device pointer handling, chunking, error flags. Host code only.
#include "f2_internal.h"

namespace {

constexpr int64_t CNN_CHUNK = 16384;  // windows per CNN launch group (activation workspace 1.75 GB, windows 92 MB)
constexpr int64_t DENSE_GROUP = 8 * CNN_CHUNK;   // windows per dense1 / dense2 launch of f2_eval_batch (conv4 + dense1 outputs: 1.3 GB)

int reset_flag(f2_ctx* ctx) {
    F2_HIP(ctx, hipMemsetAsync(ctx->flags.ptr, 0, sizeof(int), ctx->stream));
    return F2_OK;
}

// waits for the stream
int read_flag(f2_ctx* ctx, int* value) {
    F2_HIP(ctx, hipMemcpyAsync(ctx->host_flags, ctx->flags.ptr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *value = ctx->host_flags[0];
    return F2_OK;
}

int cnn_forward_device(f2_ctx* ctx, const f2_cnn* cnn, const float* d_x, int64_t n, float* d_scores, uint8_t* d_labels) {
    const size_t per = f2_cnn_workspace_floats(cnn);
    const int64_t chunk = n < CNN_CHUNK ? n : CNN_CHUNK;
    F2_TRY(f2_reserve(ctx, ctx->work, sizeof(float) * per * (size_t)chunk));
    const size_t xs = (size_t)cnn->rows * cnn->channels;
    for (int64_t s = 0; s < n; s += chunk) {
        const int64_t m = n - s < chunk ? n - s : chunk;
        F2_TRY(f2_launch_cnn(ctx, cnn, d_x + (size_t)s * xs, m, (float*)ctx->work.ptr, d_scores ? d_scores + 2 * s : nullptr,
                             d_labels ? d_labels + s : nullptr));
    }
    return F2_OK;
}

}  // namespace

extern "C" {

int f2_gather_windows(f2_ctx* ctx, const double* env, int C, int64_t N, const int64_t* centers, int64_t n_windows,
                      int radius, int step, int normalize, float* out, int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, mem_space == F2_MEM_HOST || mem_space == F2_MEM_DEVICE, F2_ERR_INVALID, "bad mem_space %d", mem_space);
    F2_CHECK(ctx, C >= 0 && N >= 0 && n_windows >= 0 && radius >= 0 && step >= 0, F2_ERR_INVALID, "negative size");
    if (n_windows == 0 || C == 0) return F2_OK;
    F2_CHECK(ctx, env && out, F2_ERR_INVALID, "null data pointer");
    const int R = 2 * radius + 1;
    const int64_t reach = (int64_t)radius * step;
    if (centers) {
        for (int64_t e = 0; e < n_windows; ++e)
            F2_CHECK(ctx, centers[e] - reach >= 0 && centers[e] + reach < N, F2_ERR_INVALID,
                     "window %lld (centre %lld, +-%lld) reaches outside the %lld-sample envelope", (long long)e,
                     (long long)centers[e], (long long)reach, (long long)N);
    } else {
        F2_CHECK(ctx, reach + (n_windows - 1) + reach < N, F2_ERR_INVALID,
                 "%lld every-sample windows do not fit in %lld samples", (long long)n_windows, (long long)N);
    }
    const int64_t* d_centers = nullptr;
    if (centers) {
        F2_TRY(f2_reserve(ctx, ctx->work2, sizeof(int64_t) * (size_t)n_windows));
        F2_TRY(f2_upload_async(ctx, ctx->work2.ptr, centers, sizeof(int64_t) * (size_t)n_windows));
        d_centers = (const int64_t*)ctx->work2.ptr;
    }
    const size_t env_bytes = sizeof(double) * (size_t)C * (size_t)N;
    const size_t out_bytes = sizeof(float) * (size_t)n_windows * R * (size_t)C;
    const double* d_env = env;
    float* d_out = out;
    if (mem_space == F2_MEM_HOST) {
        F2_TRY(f2_reserve(ctx, ctx->stage_in, env_bytes));
        F2_TRY(f2_reserve(ctx, ctx->stage_out, out_bytes));
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, env, env_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_env = (const double*)ctx->stage_in.ptr;
        d_out = (float*)ctx->stage_out.ptr;
    }
    F2_TRY(reset_flag(ctx));
    F2_TRY(f2_launch_gather(ctx, d_env, C, N, d_centers, reach, n_windows, radius, step, normalize, d_out,
                            (int*)ctx->flags.ptr));
    if (mem_space == F2_MEM_HOST)
        F2_HIP(ctx, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (normalize || mem_space == F2_MEM_HOST) {
        int flag = 0;
        F2_TRY(read_flag(ctx, &flag));
        F2_CHECK(ctx, !flag, F2_ERR_NONPOSITIVE, "values must all be positive (normalizeInput)");
    }
    return F2_OK;
}

int f2_cnn_forward(f2_ctx* ctx, const f2_cnn* cnn, const float* x, int64_t n, float* scores, uint8_t* labels,
                   int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, cnn, F2_ERR_INVALID, "cnn is NULL");
    F2_CHECK(ctx, cnn->dev == ctx->device, F2_ERR_INVALID, "cnn weights live on device %d, context on %d", cnn->dev, ctx->device);
    F2_CHECK(ctx, mem_space == F2_MEM_HOST || mem_space == F2_MEM_DEVICE, F2_ERR_INVALID, "bad mem_space %d", mem_space);
    F2_CHECK(ctx, n >= 0, F2_ERR_INVALID, "negative window count");
    if (n == 0) return F2_OK;
    F2_CHECK(ctx, x, F2_ERR_INVALID, "x is NULL");
    if (mem_space == F2_MEM_DEVICE) return cnn_forward_device(ctx, cnn, x, n, scores, labels);
    const size_t xs = (size_t)cnn->rows * cnn->channels;
    const int64_t chunk = n < CNN_CHUNK ? n : CNN_CHUNK;
    F2_TRY(f2_reserve(ctx, ctx->stage_in, sizeof(float) * xs * (size_t)chunk));
    F2_TRY(f2_reserve(ctx, ctx->stage_aux, (sizeof(float) * 2 + 1) * (size_t)chunk + 64));
    float* d_scores = (float*)ctx->stage_aux.ptr;
    uint8_t* d_labels = (uint8_t*)(d_scores + 2 * chunk);
    for (int64_t s = 0; s < n; s += chunk) {
        const int64_t m = n - s < chunk ? n - s : chunk;
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, x + (size_t)s * xs, sizeof(float) * xs * (size_t)m,
                                   hipMemcpyHostToDevice, ctx->stream));
        F2_TRY(cnn_forward_device(ctx, cnn, (const float*)ctx->stage_in.ptr, m, d_scores, d_labels));
        if (scores)
            F2_HIP(ctx, hipMemcpyAsync(scores + 2 * s, d_scores, sizeof(float) * 2 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        if (labels) F2_HIP(ctx, hipMemcpyAsync(labels + s, d_labels, (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return F2_OK;
}

int f2_eval_utterance(f2_ctx* ctx, const f2_cnn* cnn, const void* wave, int wave_dtype, int64_t N, const double* coefs,
                      int C, int lpf, double cutoff_hz, int fft_precision, int radius, int step, double* env_or_null,
                      float* scores_or_null, uint8_t* labels_or_null, int64_t* n_windows_out, int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, cnn && wave && coefs, F2_ERR_INVALID, "null argument");
    F2_CHECK(ctx, cnn->dev == ctx->device, F2_ERR_INVALID, "cnn weights live on device %d, context on %d", cnn->dev, ctx->device);
    F2_CHECK(ctx, mem_space == F2_MEM_HOST || mem_space == F2_MEM_DEVICE, F2_ERR_INVALID, "bad mem_space %d", mem_space);
    F2_CHECK(ctx, N >= 0 && C > 0 && radius >= 0 && step >= 0, F2_ERR_INVALID, "bad size");
    F2_CHECK(ctx, cnn->rows == 2 * radius + 1 && cnn->channels == C, F2_ERR_INVALID,
             "network was built for %d x %d windows, asked for %d x %d", cnn->rows, cnn->channels, 2 * radius + 1, C);
    const int R = 2 * radius + 1;
    const int64_t nb = N - (int64_t)R * step;   // Evaluating.py:73
    if (n_windows_out) *n_windows_out = nb > 0 ? nb : 0;
    if (N == 0) return F2_OK;

    // filterbank + envelope for the one utterance, all on the device
    const int64_t offsets[2] = {0, N};
    const size_t env_bytes = sizeof(double) * (size_t)C * (size_t)N;
    double* d_env;
    if (mem_space == F2_MEM_DEVICE && env_or_null) {
        d_env = env_or_null;
    } else {
        F2_TRY(f2_reserve(ctx, ctx->stage_out, env_bytes));
        d_env = (double*)ctx->stage_out.ptr;
    }
    F2_TRY(f2_upload_offsets(ctx, offsets, 1));
    F2_TRY(f2_upload_coefs(ctx, coefs, C));
    const void* d_wave = wave;
    if (mem_space == F2_MEM_HOST) {
        const size_t wb = (wave_dtype == F2_WAVE_I16 ? 2 : 8) * (size_t)N;
        F2_TRY(f2_reserve(ctx, ctx->stage_in, wb));
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, wave, wb, hipMemcpyHostToDevice, ctx->stream));
        d_wave = ctx->stage_in.ptr;
    }
    // same hand-off between the two kernels as f2_eval_batch / f2_filterbank_envelope_fused take for this length, so
    // that one utterance evaluated alone and inside a batch goes through the same envelope kernel
    f2_handoff handoff;
    F2_TRY(f2_plan_handoff(ctx, offsets, 1, C, fft_precision, false, &handoff));
    F2_TRY(f2_launch_filterbank(ctx, d_wave, wave_dtype, (const int64_t*)ctx->offsets.ptr, offsets,
                                (const double*)ctx->coefs.ptr, 1, C, d_env, &handoff));
    F2_TRY(f2_launch_envelope(ctx, d_env, (const int64_t*)ctx->offsets.ptr, offsets, 1, C, lpf, cutoff_hz, fft_precision,
                              d_env, &handoff));
    if (mem_space == F2_MEM_HOST && env_or_null)
        F2_HIP(ctx, hipMemcpyAsync(env_or_null, d_env, env_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (nb <= 0) {
        if (mem_space == F2_MEM_HOST) F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return F2_OK;
    }
    // every-sample windows -> normalise -> CNN, chunk by chunk; nothing leaves HBM
    const int64_t chunk = nb < CNN_CHUNK ? nb : CNN_CHUNK;
    F2_TRY(f2_reserve(ctx, ctx->xbuf, sizeof(float) * (size_t)chunk * R * (size_t)C));
    F2_TRY(f2_reserve(ctx, ctx->work, sizeof(float) * f2_cnn_workspace_floats(cnn) * (size_t)chunk));
    float* d_scores = scores_or_null;
    uint8_t* d_labels = labels_or_null;
    if (mem_space == F2_MEM_HOST) {
        F2_TRY(f2_reserve(ctx, ctx->stage_aux, (sizeof(float) * 2 + 1) * (size_t)nb + 64));
        d_scores = (float*)ctx->stage_aux.ptr;
        d_labels = (uint8_t*)(d_scores + 2 * nb);
    }
    F2_TRY(reset_flag(ctx));
    const int64_t reach = (int64_t)radius * step;
    for (int64_t s = 0; s < nb; s += chunk) {
        const int64_t m = nb - s < chunk ? nb - s : chunk;
        F2_TRY(f2_launch_gather(ctx, d_env, C, N, nullptr, reach + s, m, radius, step, 1, (float*)ctx->xbuf.ptr,
                                (int*)ctx->flags.ptr));
        F2_TRY(f2_launch_cnn(ctx, cnn, (const float*)ctx->xbuf.ptr, m, (float*)ctx->work.ptr,
                             d_scores ? d_scores + 2 * s : nullptr, d_labels ? d_labels + s : nullptr));
    }
    if (mem_space == F2_MEM_HOST) {
        if (scores_or_null)
            F2_HIP(ctx, hipMemcpyAsync(scores_or_null, d_scores, sizeof(float) * 2 * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
        if (labels_or_null)
            F2_HIP(ctx, hipMemcpyAsync(labels_or_null, d_labels, (size_t)nb, hipMemcpyDeviceToHost, ctx->stream));
    }
    int flag = 0;
    F2_TRY(read_flag(ctx, &flag));
    F2_CHECK(ctx, !flag, F2_ERR_NONPOSITIVE, "values must all be positive (normalizeInput)");
    return F2_OK;
}

int f2_eval_batch(f2_ctx* ctx, const f2_cnn* cnn, const void* wave, int wave_dtype, const int64_t* offsets,
                  const double* coefs, int B, int C, int lpf, double cutoff_hz, int fft_precision, int radius, int step,
                  float* scores_or_null, uint8_t* labels_or_null, int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, cnn && coefs && offsets, F2_ERR_INVALID, "null argument");
    F2_CHECK(ctx, cnn->dev == ctx->device, F2_ERR_INVALID, "cnn weights live on device %d, context on %d", cnn->dev, ctx->device);
    F2_CHECK(ctx, mem_space == F2_MEM_HOST || mem_space == F2_MEM_DEVICE, F2_ERR_INVALID, "bad mem_space %d", mem_space);
    F2_CHECK(ctx, wave_dtype == F2_WAVE_I16 || wave_dtype == F2_WAVE_F64, F2_ERR_INVALID, "bad wave_dtype %d", wave_dtype);
    F2_CHECK(ctx, fft_precision == F2_FFT_F32 || fft_precision == F2_FFT_F64, F2_ERR_INVALID, "bad fft_precision %d", fft_precision);
    F2_CHECK(ctx, !lpf || (cutoff_hz > 0 && cutoff_hz < 8000), F2_ERR_INVALID, "cutoff %g Hz outside (0, 8000)", cutoff_hz);
    F2_CHECK(ctx, B >= 0 && C > 0 && radius >= 0 && step >= 0, F2_ERR_INVALID, "bad size");
    F2_CHECK(ctx, cnn->rows == 2 * radius + 1 && cnn->channels == C, F2_ERR_INVALID,
             "network was built for %d x %d windows, asked for %d x %d", cnn->rows, cnn->channels, 2 * radius + 1, C);
    F2_CHECK(ctx, offsets[0] == 0, F2_ERR_INVALID, "offsets[0] must be 0");
    for (int b = 0; b < B; ++b)
        F2_CHECK(ctx, offsets[b + 1] >= offsets[b], F2_ERR_INVALID, "offsets must not decrease (utterance %d)", b);
    const int64_t total = B > 0 ? offsets[B] : 0;
    if (total == 0) return F2_OK;
    F2_CHECK(ctx, wave, F2_ERR_INVALID, "null wave");
    const int R = 2 * radius + 1;
    int64_t nb_total = 0, nb_max = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t nb = offsets[b + 1] - offsets[b] - (int64_t)R * step;   // Evaluating.py:73
        if (nb > 0) {
            nb_total += nb;
            nb_max = nb > nb_max ? nb : nb_max;
        }
    }

    // filterbank + envelope of the whole batch
    F2_TRY(f2_upload_offsets(ctx, offsets, B));
    F2_TRY(f2_upload_coefs(ctx, coefs, C));
    F2_TRY(f2_reserve(ctx, ctx->stage_out, sizeof(double) * (size_t)C * (size_t)total));
    double* d_env = (double*)ctx->stage_out.ptr;
    const void* d_wave = wave;
    if (mem_space == F2_MEM_HOST) {
        const size_t wb = (wave_dtype == F2_WAVE_I16 ? 2 : 8) * (size_t)total;
        F2_TRY(f2_reserve(ctx, ctx->stage_in, wb));
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, wave, wb, hipMemcpyHostToDevice, ctx->stream));
        d_wave = ctx->stage_in.ptr;
    }
    f2_handoff handoff;
    F2_TRY(f2_plan_handoff(ctx, offsets, B, C, fft_precision, false, &handoff));
    F2_TRY(f2_launch_filterbank(ctx, d_wave, wave_dtype, (const int64_t*)ctx->offsets.ptr, offsets,
                                (const double*)ctx->coefs.ptr, B, C, d_env, &handoff));
    F2_TRY(f2_launch_envelope(ctx, d_env, (const int64_t*)ctx->offsets.ptr, offsets, B, C, lpf, cutoff_hz, fft_precision,
                              d_env, &handoff));
    if (nb_total == 0) {
        if (mem_space == F2_MEM_HOST) F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return F2_OK;
    }

    // every-sample windows -> normalise -> conv1 .. conv4, utterance by utterance, chunk by chunk; the dense layers run over the
    // conv4 outputs of up to DENSE_GROUP windows at once (dense1's grid of 64-window workgroups then fills whole rounds of the
    // device: launched per 14 240-window utterance its second round was one third full); nothing leaves HBM
    const int64_t chunk = nb_max < CNN_CHUNK ? nb_max : CNN_CHUNK;
    const int64_t group_cap = nb_total < DENSE_GROUP ? nb_total : DENSE_GROUP;
    const size_t conv_floats = f2_cnn_workspace_floats(cnn) - f2_cnn_dense_floats(cnn), flat = f2_cnn_flat_floats(cnn);
    F2_TRY(f2_reserve(ctx, ctx->xbuf, sizeof(float) * (size_t)chunk * R * (size_t)C));
    F2_TRY(f2_reserve(ctx, ctx->work, sizeof(float) * conv_floats * (size_t)chunk));
    F2_TRY(f2_reserve(ctx, ctx->dense_in, sizeof(float) * f2_cnn_dense_floats(cnn) * (size_t)group_cap));
    float* const d_a4 = (float*)ctx->dense_in.ptr;
    float* const d_a5 = d_a4 + flat * (size_t)group_cap;
    float* d_scores = scores_or_null;
    uint8_t* d_labels = labels_or_null;
    if (mem_space == F2_MEM_HOST) {
        F2_TRY(f2_reserve(ctx, ctx->stage_aux, (sizeof(float) * 2 + 1) * (size_t)nb_total + 64));
        d_scores = (float*)ctx->stage_aux.ptr;
        d_labels = (uint8_t*)(d_scores + 2 * nb_total);
    }
    F2_TRY(reset_flag(ctx));
    const int64_t reach = (int64_t)radius * step;
    int64_t done = 0, g0 = 0, gn = 0;      // windows finished before this utterance; first window and size of the open dense group
    auto flush = [&]() -> int {
        if (gn > 0)
            F2_TRY(f2_launch_cnn_dense(ctx, cnn, d_a4, gn, d_a5, d_scores ? d_scores + 2 * g0 : nullptr, d_labels ? d_labels + g0 : nullptr));
        g0 += gn;
        gn = 0;
        return F2_OK;
    };
    for (int b = 0; b < B; ++b) {
        const int64_t N = offsets[b + 1] - offsets[b];
        const int64_t nb = N - (int64_t)R * step;
        if (nb <= 0) continue;
        const double* env_b = d_env + (size_t)C * (size_t)offsets[b];
        for (int64_t s = 0; s < nb; s += chunk) {
            const int64_t m = nb - s < chunk ? nb - s : chunk;
            if (gn + m > group_cap) F2_TRY(flush());
            F2_TRY(f2_launch_gather(ctx, env_b, C, N, nullptr, reach + s, m, radius, step, 1, (float*)ctx->xbuf.ptr,
                                    (int*)ctx->flags.ptr));
            F2_TRY(f2_launch_cnn_convs(ctx, cnn, (const float*)ctx->xbuf.ptr, m, (float*)ctx->work.ptr, d_a4 + flat * (size_t)gn));
            gn += m;
        }
        done += nb;
    }
    F2_TRY(flush());
    if (mem_space == F2_MEM_HOST) {
        if (scores_or_null)
            F2_HIP(ctx, hipMemcpyAsync(scores_or_null, d_scores, sizeof(float) * 2 * (size_t)nb_total, hipMemcpyDeviceToHost, ctx->stream));
        if (labels_or_null)
            F2_HIP(ctx, hipMemcpyAsync(labels_or_null, d_labels, (size_t)nb_total, hipMemcpyDeviceToHost, ctx->stream));
    }
    int flag = 0;
    F2_TRY(read_flag(ctx, &flag));
    F2_CHECK(ctx, !flag, F2_ERR_NONPOSITIVE, "values must all be positive (normalizeInput)");
    return F2_OK;
}

}  // extern "C"
