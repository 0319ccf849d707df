#include "f2_internal.h"
extern "C" {
int f2_gather_windows(f2_ctx* ctx, const double*, int, int64_t, const int64_t*, int64_t, int, int, int, float*, int) { return f2_fail(ctx, F2_ERR_UNSUPPORTED, "nyi"); }
int f2_cnn_create(f2_ctx* ctx, const float* const*, int, int, f2_cnn**) { return f2_fail(ctx, F2_ERR_UNSUPPORTED, "nyi"); }
int f2_cnn_destroy(f2_ctx*, f2_cnn*) { return 0; }
int f2_cnn_forward(f2_ctx* ctx, const f2_cnn*, const float*, int64_t, float*, uint8_t*, int) { return f2_fail(ctx, F2_ERR_UNSUPPORTED, "nyi"); }
int f2_eval_utterance(f2_ctx* ctx, const f2_cnn*, const void*, int, int64_t, const double*, int, int, double, int, int, int, double*, float*, uint8_t*, int64_t*, int) { return f2_fail(ctx, F2_ERR_UNSUPPORTED, "nyi"); }
}
