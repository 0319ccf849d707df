// C-ABI entry points of libf2cnn_hip.so: context, memory/timing helpers, and the host/device
// pointer handling around the kernel launchers. See include/f2cnn_hip.h for the contract.
#include "f2_internal.h"

char g_f2_err[512] = {0};

int f2_fail(f2_ctx* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->err : g_f2_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

int f2_reserve(f2_ctx* ctx, f2_scratch& s, size_t bytes) {
    if (bytes <= s.bytes) return F2_OK;
    if (s.ptr) {
        // the old block may still be in use by work queued on the stream
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        F2_HIP(ctx, hipFree(s.ptr));
        s.ptr = nullptr;
        s.bytes = 0;
    }
    size_t want = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&s.ptr, want);
    if (e != hipSuccess) {
        s.ptr = nullptr;
        return f2_fail(ctx, F2_ERR_NOMEM, "hipMalloc(%zu) -> %s", want, hipGetErrorString(e));
    }
    s.bytes = want;
    return F2_OK;
}

int f2_upload_async(f2_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    if (bytes == 0) return F2_OK;
    constexpr size_t RING = size_t(8) << 20, ALIGN = 256;
    if (bytes > RING / 4) {
        // Large uploads (the `centers` array of f2_gather_windows, twiddle tables) go through a few grow-only page-locked
        // side buffers that are reused once the copy that read them has finished: a hipHostMalloc / hipHostFree pair per
        // call is slow and can serialise with the device (round-4 advisor finding).
        f2_ctx::up_side* pick = nullptr;   // the smallest idle buffer that is large enough, else the smallest idle one (to be grown)
        for (f2_ctx::up_side& b : ctx->up_big) {
            if (b.busy && hipEventQuery(b.done) == hipSuccess) b.busy = false;
            if (b.busy) continue;
            const bool fits = b.cap >= bytes, cur_fits = pick && pick->cap >= bytes;
            if (!pick || (fits && !cur_fits) || (fits == cur_fits && b.cap < pick->cap)) pick = &b;
        }
        if (!pick && ctx->up_big.size() >= 4) {   // all busy: wait for the oldest rather than pinning ever more memory
            pick = &ctx->up_big.front();
            F2_HIP(ctx, hipEventSynchronize(pick->done));
            pick->busy = false;
        }
        if (!pick) {
            hipEvent_t ev;
            F2_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->up_big.push_back({nullptr, 0, ev, false});
            pick = &ctx->up_big.back();
        }
        if (pick->cap < bytes) {
            if (pick->ptr) (void)hipHostFree(pick->ptr);
            pick->ptr = nullptr;
            pick->cap = 0;
            const size_t want = std::max(bytes + bytes / 4, size_t(4) << 20);
            hipError_t e = hipHostMalloc(&pick->ptr, want, hipHostMallocDefault);
            if (e != hipSuccess) {
                pick->ptr = nullptr;
                return f2_fail(ctx, F2_ERR_NOMEM, "hipHostMalloc(%zu) -> %s", want, hipGetErrorString(e));
            }
            pick->cap = want;
        }
        memcpy(pick->ptr, src, bytes);
        F2_HIP(ctx, hipMemcpyAsync(d_dst, pick->ptr, bytes, hipMemcpyHostToDevice, ctx->stream));
        pick->busy = true;     // (set before the record: if that fails the buffer is waited for by event query -> stays busy
        F2_HIP(ctx, hipEventRecord(pick->done, ctx->stream));   //  until the slow path above synchronises on it)
        return F2_OK;
    }
    hipEvent_t ev;
    if (!ctx->prof_pool.empty()) {
        ev = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
    } else {
        F2_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));   // (shares the pool of the timing events: the device only READS the guarded buffer)
    }
    // (from here on an early return hands `ev` back to the pool)
    struct ev_guard {
        f2_ctx* c;
        hipEvent_t e;
        bool armed;
        ~ev_guard() {
            if (armed) c->prof_pool.push_back(e);
        }
    } guard{ctx, ev, true};
    if (!ctx->up_ring) {
        F2_HIP(ctx, hipHostMalloc((void**)&ctx->up_ring, RING, hipHostMallocDefault));
        ctx->up_cap = RING;
    }
    const size_t need = (bytes + ALIGN - 1) & ~(ALIGN - 1);
    size_t begin = ctx->up_head;
    if (begin + need > ctx->up_cap) begin = 0;
    // the ring region [begin, begin + need) must not be the source of a copy that is still pending (oldest spans first;
    // normally long done: the ring holds thousands of the small arrays a call uploads)
    while (!ctx->up_inflight.empty()) {
        const f2_ctx::up_span& sp = ctx->up_inflight.front();
        const bool overlaps = sp.begin < begin + need && begin < sp.end;
        if (!overlaps && hipEventQuery(sp.done) != hipSuccess) break;
        if (overlaps) F2_HIP(ctx, hipEventSynchronize(sp.done));
        ctx->prof_pool.push_back(sp.done);
        ctx->up_inflight.erase(ctx->up_inflight.begin());
    }
    for (const f2_ctx::up_span& sp : ctx->up_inflight)
        if (sp.begin < begin + need && begin < sp.end) F2_HIP(ctx, hipEventSynchronize(sp.done));
    memcpy(ctx->up_ring + begin, src, bytes);
    F2_HIP(ctx, hipMemcpyAsync(d_dst, ctx->up_ring + begin, bytes, hipMemcpyHostToDevice, ctx->stream));
    F2_HIP(ctx, hipEventRecord(ev, ctx->stream));
    guard.armed = false;
    ctx->up_inflight.push_back({begin, begin + need, ev});
    ctx->up_head = begin + need;
    return F2_OK;
}

static int prof_event(f2_ctx* ctx, hipEvent_t* ev) {
    if (!ctx->prof_pool.empty()) {
        *ev = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return F2_OK;
    }
    // (timing events only: without the system-scope fence - a cache write-back of ~10 us between two kernels - that a default
    // event performs when it is reached; the events of f2_event_create, which order host reads after device writes, keep it)
    F2_HIP(ctx, hipEventCreateWithFlags(ev, hipEventDisableSystemFence));
    return F2_OK;
}

int f2_prof_begin(f2_ctx* ctx, int kernel_id) {
    if (!ctx->prof_on) return F2_OK;
    auto& spans = ctx->prof[kernel_id];
    if (!spans.empty() && !spans.back().closed) {   // left open by a launch that returned an error
        ctx->prof_pool.push_back(spans.back().first);
        ctx->prof_pool.push_back(spans.back().second);
        spans.pop_back();
    }
    hipEvent_t a, b;
    F2_TRY(prof_event(ctx, &a));
    F2_TRY(prof_event(ctx, &b));
    spans.push_back({a, b, false});
    F2_HIP(ctx, hipEventRecord(a, ctx->stream));
    return F2_OK;
}

int f2_prof_end(f2_ctx* ctx, int kernel_id) {
    if (!ctx->prof_on || ctx->prof[kernel_id].empty()) return F2_OK;
    F2_HIP(ctx, hipEventRecord(ctx->prof[kernel_id].back().second, ctx->stream));
    ctx->prof[kernel_id].back().closed = true;
    return F2_OK;
}

extern "C" {

int f2_version(void) { return 105; }   // 101: f2_eval_batch; 102: f2_host_alloc, F2_MEM_HOST_ASYNC; 103: f2_ctx_set_option; 104: f2_event_query; 105: f2_spectral_guard_read

int f2_device_count(int* count) {
    if (!count) return f2_fail(nullptr, F2_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return f2_fail(nullptr, F2_ERR_HIP, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    *count = n;
    return F2_OK;
}

int f2_ctx_create(int device, f2_ctx** out) {
    if (!out) return f2_fail(nullptr, F2_ERR_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return f2_fail(nullptr, F2_ERR_HIP, "no HIP device available (%s)",
                       e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return f2_fail(nullptr, F2_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    f2_ctx* ctx = new f2_ctx();
    ctx->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->host_flags, 64, hipHostMallocDefault);
    if (e != hipSuccess) {
        f2_fail(nullptr, F2_ERR_HIP, "context setup on device %d -> %s", device, hipGetErrorString(e));
        delete ctx;
        return F2_ERR_HIP;
    }
    ctx->num_cus = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        f2_fail(nullptr, F2_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
        (void)hipStreamDestroy(ctx->stream);
        (void)hipHostFree(ctx->host_flags);
        delete ctx;
        return F2_ERR_UNSUPPORTED;
    }
    int rc = f2_reserve(ctx, ctx->flags, 64);
    if (rc != F2_OK) {
        memcpy(g_f2_err, ctx->err, sizeof(g_f2_err));
        (void)hipStreamDestroy(ctx->stream);
        (void)hipHostFree(ctx->host_flags);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return F2_OK;
}

int f2_ctx_destroy(f2_ctx* ctx) {
    if (!ctx) return F2_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    f2_scratch* all[] = {&ctx->coefs, &ctx->offsets, &ctx->stage_in, &ctx->stage_out, &ctx->stage_aux,
                         &ctx->work,  &ctx->work2,   &ctx->xbuf,      &ctx->flags,    &ctx->gather_log, &ctx->dense_in};
    for (f2_scratch* s : all)
        if (s->ptr) (void)hipFree(s->ptr);
    for (auto& v : ctx->prof)
        for (auto& pr : v) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    for (auto& prec : ctx->tw)
        for (f2_scratch& s : prec)
            if (s.ptr) (void)hipFree(s.ptr);
    for (auto& prec : ctx->tw_large)
        for (f2_scratch& s : prec)
            if (s.ptr) (void)hipFree(s.ptr);
    for (f2_scratch& s : ctx->tw_p3)
        if (s.ptr) (void)hipFree(s.ptr);
    for (f2_scratch& s : ctx->tw_fl)
        if (s.ptr) (void)hipFree(s.ptr);
    for (f2_scratch& s : ctx->tw_split)
        if (s.ptr) (void)hipFree(s.ptr);
    for (f2_scratch* sc : {&ctx->tw_pair[0], &ctx->tw_pair[1], &ctx->pair_list[0], &ctx->pair_list[1]})
        if (sc->ptr) (void)hipFree(sc->ptr);
    if (ctx->work3.ptr) (void)hipFree(ctx->work3.ptr);
    if (ctx->k1_states.ptr) (void)hipFree(ctx->k1_states.ptr);
    if (ctx->k1_mtab.ptr) (void)hipFree(ctx->k1_mtab.ptr);
    if (ctx->k1_order.ptr) (void)hipFree(ctx->k1_order.ptr);
    if (ctx->handoff.ptr) (void)hipFree(ctx->handoff.ptr);
    if (ctx->handoff_off.ptr) (void)hipFree(ctx->handoff_off.ptr);
    for (auto& t : ctx->spec_tabs)
        for (f2_scratch* sc : {&t.hu, &t.e, &t.lgroup, &t.e64})
            if (sc->ptr) (void)hipFree(sc->ptr);
    for (f2_scratch* sc : {&ctx->spec_x, &ctx->spec_rho, &ctx->spec_xpart, &ctx->spec_meta, &ctx->spec_uflag, &ctx->spec_lptab})
        if (sc->ptr) (void)hipFree(sc->ptr);
    for (auto& prec : ctx->tw_sp)
        for (f2_scratch& sc : prec)
            if (sc.ptr) (void)hipFree(sc.ptr);
    for (auto& sp : ctx->up_inflight) (void)hipEventDestroy(sp.done);
    for (auto& b : ctx->up_big) {
        if (b.ptr) (void)hipHostFree(b.ptr);
        (void)hipEventDestroy(b.done);
    }
    if (ctx->up_ring) (void)hipHostFree(ctx->up_ring);
    if (ctx->host_flags) (void)hipHostFree(ctx->host_flags);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return F2_OK;
}

int f2_ctx_synchronize(f2_ctx* ctx) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F2_OK;
}

int f2_ctx_set_stream(f2_ctx* ctx, void* hip_stream) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) F2_HIP(ctx, hipStreamDestroy(ctx->stream));
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        F2_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return F2_OK;
}

void* f2_ctx_get_stream(f2_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

const char* f2_last_error(f2_ctx* ctx) { return ctx ? ctx->err : g_f2_err; }

namespace {
struct opt_entry {
    const char* key;
    int f2_ctx::*ifield;
    float f2_ctx::*ffield;
    double lo, hi;
};
const opt_entry kOptions[] = {
    {"spectral", &f2_ctx::opt_spectral, nullptr, 0, 1},
    {"spectral_tol", nullptr, &f2_ctx::opt_spectral_tol, 0, 1},
    {"spectral_min_rows", &f2_ctx::opt_spectral_min_rows, nullptr, 0, 1 << 30},
    {"spectral_min_pad", &f2_ctx::opt_spectral_min_pad, nullptr, -1, 1 << 16},
    {"spectral_guard_dump", &f2_ctx::opt_spectral_guard_dump, nullptr, 0, 1},
    {"k1_split", &f2_ctx::opt_k1_split, nullptr, -1, 64},
    {"k1_queue", &f2_ctx::opt_k1_queue, nullptr, -1, 1},
    {"k1_qwaves", &f2_ctx::opt_k1_qwaves, nullptr, 0, 1 << 20},
    {"env_pair", &f2_ctx::opt_env_pair, nullptr, 0, 1},
    {"env_plan4", &f2_ctx::opt_env_plan4, nullptr, 0, 1},
    {"cnn_f16x3", &f2_ctx::opt_cnn_bf16x3, nullptr, 0, 1},
    {"cnn_bf16x3", &f2_ctx::opt_cnn_bf16x3, nullptr, 0, 1},   // the option's name in rounds 3-4 (bf16 pieces then): same switch
    {"cnn_ws", &f2_ctx::opt_cnn_ws, nullptr, 0, 1},
    {"cnn_ws_dense", &f2_ctx::opt_cnn_ws_dense, nullptr, 0, 1},
    {"gather_blocked", &f2_ctx::opt_gather_blocked, nullptr, 0, 1},
};
const opt_entry* find_option(const char* key) {
    if (!key) return nullptr;
    for (const opt_entry& e : kOptions)
        if (strcmp(e.key, key) == 0) return &e;
    return nullptr;
}
}  // namespace

int f2_ctx_set_option(f2_ctx* ctx, const char* key, double value) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    const opt_entry* e = find_option(key);
    F2_CHECK(ctx, e, F2_ERR_INVALID, "unknown option '%s'", key ? key : "(null)");
    F2_CHECK(ctx, value >= e->lo && value <= e->hi, F2_ERR_INVALID, "option %s: %g outside [%g, %g]", key, value, e->lo, e->hi);
    if (e->ifield) {
        F2_CHECK(ctx, value == (double)(int)value, F2_ERR_INVALID, "option %s takes an integer", key);
        ctx->*(e->ifield) = (int)value;
    } else {
        ctx->*(e->ffield) = (float)value;
    }
    return F2_OK;
}

int f2_ctx_get_option(f2_ctx* ctx, const char* key, double* value) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (key && value && strncmp(key, "spectral_routed", 15) == 0) {
        // read-only: utterances (or, "..._samples", their samples) the last fused call sent through the spectral kernel
        const bool samples = strcmp(key, "spectral_routed_samples") == 0;
        F2_CHECK(ctx, samples || strcmp(key, "spectral_routed") == 0, F2_ERR_INVALID, "unknown option '%s'", key);
        *value = 0;
        const size_t B = ctx->spec_last_B;
        if (B == 0 || ctx->spec_meta_host.size() < B || ctx->offsets_host.size() != B + 1) return F2_OK;
        double acc = 0;
        for (size_t b = 0; b < B; ++b)
            if (ctx->spec_meta_host[b] == 0) acc += samples ? (double)(ctx->offsets_host[b + 1] - ctx->offsets_host[b]) : 1.0;
        *value = acc;
        return F2_OK;
    }
    if (key && value && strncmp(key, "spectral_flagged", 16) == 0) {
        // read-only: utterances ("..._samples": their samples) of the last fused call that the spectral kernel's accuracy
        // guard sent back to the filterbank kernel + envelope kernel (waits for the stream)
        const bool samples = strcmp(key, "spectral_flagged_samples") == 0;
        F2_CHECK(ctx, samples || strcmp(key, "spectral_flagged") == 0, F2_ERR_INVALID, "unknown option '%s'", key);
        *value = 0;
        const size_t B = ctx->spec_last_B;
        if (B == 0 || !ctx->spec_uflag.ptr || ctx->spec_meta_host.size() < B) return F2_OK;
        std::vector<int> flags(B);
        F2_HIP(ctx, hipMemcpyAsync(flags.data(), ctx->spec_uflag.ptr, sizeof(int) * B, hipMemcpyDeviceToHost, ctx->stream));
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double acc = 0;
        const bool have_off = ctx->offsets_host.size() == B + 1;
        for (size_t b = 0; b < B; ++b)
            if (flags[b] != 0 && ctx->spec_meta_host[b] == 0)
                acc += samples ? (have_off ? (double)(ctx->offsets_host[b + 1] - ctx->offsets_host[b]) : 0.0) : 1.0;
        *value = acc;
        return F2_OK;
    }
    const opt_entry* e = find_option(key);
    F2_CHECK(ctx, e && value, F2_ERR_INVALID, "unknown option '%s'", key ? key : "(null)");
    *value = e->ifield ? (double)(ctx->*(e->ifield)) : (double)(ctx->*(e->ffield));
    return F2_OK;
}

int f2_spectral_guard_read(f2_ctx* ctx, float* out, int64_t rows, int64_t* rows_available) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, rows >= 0 && (out || rows == 0), F2_ERR_INVALID, "bad output buffer");
    if (rows_available) *rows_available = (int64_t)ctx->spec_gdump_rows;
    const size_t take = std::min((size_t)rows, ctx->spec_gdump_rows);
    if (take == 0) return F2_OK;
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_HIP(ctx, hipMemcpyAsync(out, ctx->spec_gdump.ptr, sizeof(float) * 4 * take, hipMemcpyDeviceToHost, ctx->stream));
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F2_OK;
}

int f2_dev_malloc(f2_ctx* ctx, size_t bytes, void** dptr) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, dptr, F2_ERR_INVALID, "dptr is NULL");
    *dptr = nullptr;
    if (bytes == 0) return F2_OK;
    F2_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return f2_fail(ctx, F2_ERR_NOMEM, "hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    return F2_OK;
}

int f2_dev_free(f2_ctx* ctx, void* dptr) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (!dptr) return F2_OK;
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    F2_HIP(ctx, hipFree(dptr));
    return F2_OK;
}

int f2_host_alloc(f2_ctx* ctx, size_t bytes, void** hptr) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, hptr, F2_ERR_INVALID, "hptr is NULL");
    *hptr = nullptr;
    if (bytes == 0) return F2_OK;
    F2_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(hptr, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        *hptr = nullptr;
        return f2_fail(ctx, F2_ERR_NOMEM, "hipHostMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    }
    return F2_OK;
}

int f2_host_free(f2_ctx* ctx, void* hptr) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (!hptr) return F2_OK;
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    F2_HIP(ctx, hipHostFree(hptr));
    return F2_OK;
}

int f2_dev_memset(f2_ctx* ctx, void* dptr, int value, size_t bytes) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (bytes == 0) return F2_OK;
    F2_CHECK(ctx, dptr, F2_ERR_INVALID, "dptr is NULL");
    F2_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return F2_OK;
}

int f2_memcpy_h2d(f2_ctx* ctx, void* dst, const void* src, size_t bytes) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (bytes == 0) return F2_OK;
    F2_CHECK(ctx, dst && src, F2_ERR_INVALID, "null pointer in h2d copy");
    F2_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F2_OK;
}

int f2_memcpy_d2h(f2_ctx* ctx, void* dst, const void* src, size_t bytes) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (bytes == 0) return F2_OK;
    F2_CHECK(ctx, dst && src, F2_ERR_INVALID, "null pointer in d2h copy");
    F2_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F2_OK;
}

int f2_event_create(f2_ctx* ctx, void** event) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, event, F2_ERR_INVALID, "event out pointer is NULL");
    hipEvent_t ev;
    F2_HIP(ctx, hipEventCreate(&ev));
    *event = (void*)ev;
    return F2_OK;
}

int f2_event_destroy(f2_ctx* ctx, void* event) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    if (event) F2_HIP(ctx, hipEventDestroy((hipEvent_t)event));
    return F2_OK;
}

int f2_event_record(f2_ctx* ctx, void* event) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, event, F2_ERR_INVALID, "event is NULL");
    F2_HIP(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return F2_OK;
}

int f2_event_elapsed_ms(f2_ctx* ctx, void* start, void* stop, float* ms) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, start && stop && ms, F2_ERR_INVALID, "null argument");
    F2_HIP(ctx, hipEventSynchronize((hipEvent_t)stop));
    F2_HIP(ctx, hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return F2_OK;
}

int f2_event_query(f2_ctx* ctx, void* event, int* done) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, event && done, F2_ERR_INVALID, "null argument");
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e != hipSuccess && e != hipErrorNotReady) return f2_fail(ctx, F2_ERR_HIP, "hipEventQuery -> %s", hipGetErrorString(e));
    *done = e == hipSuccess ? 1 : 0;
    return F2_OK;
}

int f2_prof_enable(f2_ctx* ctx, int on) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    ctx->prof_on = on != 0;
    return on ? f2_prof_reset(ctx) : F2_OK;
}

int f2_prof_reset(f2_ctx* ctx) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& v : ctx->prof) {
        for (auto& pr : v) {
            ctx->prof_pool.push_back(pr.first);
            ctx->prof_pool.push_back(pr.second);
        }
        v.clear();
    }
    return F2_OK;
}

int f2_prof_get(f2_ctx* ctx, int kernel_id, int* launches, float* total_ms) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_CHECK(ctx, kernel_id >= 0 && kernel_id < F2_K_COUNT, F2_ERR_INVALID, "bad kernel id %d", kernel_id);
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float sum = 0.f;
    int closed = 0;
    for (auto& pr : ctx->prof[kernel_id]) {
        if (!pr.closed) continue;
        float ms = 0.f;
        F2_HIP(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        sum += ms;
        ++closed;
    }
    if (launches) *launches = closed;
    if (total_ms) *total_ms = sum;
    return F2_OK;
}

const char* f2_prof_kernel_name(int kernel_id) {
    static const char* names[F2_K_COUNT] = {"k_erb_filterbank", "k_envelope", "k_gather_windows", "k_cnn_forward",
                                            "k_spectral_envelope", "k_utterance_spectrum", "k_tail_state"};
    return kernel_id >= 0 && kernel_id < F2_K_COUNT ? names[kernel_id] : "";
}

// ------------------------------------------------------------------------------------------------
// argument checking shared by the batched DSP entry points
// ------------------------------------------------------------------------------------------------
static int check_batch(f2_ctx* ctx, const int64_t* offsets, int B, int C, int mem_space) {
    F2_CHECK(ctx, B >= 0 && C >= 0, F2_ERR_INVALID, "negative batch (B=%d) or channel count (C=%d)", B, C);
    F2_CHECK(ctx, mem_space == F2_MEM_HOST || mem_space == F2_MEM_DEVICE || mem_space == F2_MEM_HOST_ASYNC, F2_ERR_INVALID,
             "bad mem_space %d", mem_space);
    F2_CHECK(ctx, offsets, F2_ERR_INVALID, "offsets is NULL");
    F2_CHECK(ctx, offsets[0] == 0, F2_ERR_INVALID, "offsets[0] must be 0");
    for (int b = 0; b < B; ++b)
        F2_CHECK(ctx, offsets[b + 1] >= offsets[b], F2_ERR_INVALID, "offsets must be non-decreasing (b=%d)", b);
    return F2_OK;
}

}  // extern "C" (internal helpers follow)

int f2_upload_offsets(f2_ctx* ctx, const int64_t* offsets, int B) {
    if (ctx->offsets_host.size() == (size_t)(B + 1) &&
        memcmp(ctx->offsets_host.data(), offsets, sizeof(int64_t) * (size_t)(B + 1)) == 0)
        return F2_OK;  // same batch shape as the previous call: the device copy is still valid
    ctx->offsets_host.clear();
    F2_TRY(f2_reserve(ctx, ctx->offsets, sizeof(int64_t) * (size_t)(B + 1)));
    // (the host array belongs to the caller: staged through page-locked memory, the call does not wait for the copy)
    F2_TRY(f2_upload_async(ctx, ctx->offsets.ptr, offsets, sizeof(int64_t) * (size_t)(B + 1)));
    ctx->offsets_host.assign(offsets, offsets + B + 1);
    return F2_OK;
}

int f2_upload_coefs(f2_ctx* ctx, const double* coefs, int C) {
    if (ctx->coefs_host.size() == (size_t)C * 10 &&
        memcmp(ctx->coefs_host.data(), coefs, sizeof(double) * 10 * (size_t)C) == 0)
        return F2_OK;
    ctx->coefs_host.clear();
    F2_TRY(f2_reserve(ctx, ctx->coefs, sizeof(double) * 10 * (size_t)C));
    F2_TRY(f2_upload_async(ctx, ctx->coefs.ptr, coefs, sizeof(double) * 10 * (size_t)C));
    ctx->coefs_host.assign(coefs, coefs + (size_t)C * 10);
    ctx->spec_coefs_ok = -1;      // eligibility of this table for the spectral kernel: decided on first use
    return F2_OK;
}

int f2_plan_handoff(f2_ctx* ctx, const int64_t* h_offsets, int B, int C, int precision, bool want_gfb, f2_handoff* plan) {
    *plan = f2_handoff();
    if (want_gfb || precision != F2_FFT_F32) return F2_OK;
    constexpr size_t SCRATCH_LIMIT = size_t(64) << 30;
    std::vector<int64_t> off((size_t)B, -1);
    int64_t floats = 0;
    bool any_long = false;
    for (int b = 0; b < B; ++b) {
        const int64_t n = h_offsets[b + 1] - h_offsets[b];
        // rows of the single-row LDS-resident kernels hand over inside their own output slot; the two-sub-row kernel
        // (32769..65536 samples) and the four-step path read compact scratch rows
        if (n <= (int64_t(1) << 15)) continue;
        const int log2h = f2_log2_ceil(n) - 1;
        if (!f2_envelope_pair_supports(log2h, precision) && !f2_envelope_split_supports(log2h, precision))
            return F2_OK;   // a row for the general path: float64 for all
        off[(size_t)b] = floats;
        floats += (int64_t)C * n;
        any_long = true;
    }
    if ((size_t)floats * sizeof(float) > SCRATCH_LIMIT) return F2_OK;
    plan->f32 = true;
    if (!any_long) return F2_OK;
    F2_TRY(f2_reserve(ctx, ctx->handoff, sizeof(float) * (size_t)floats));
    F2_TRY(f2_reserve(ctx, ctx->handoff_off, sizeof(int64_t) * (size_t)B));
    F2_TRY(f2_upload_async(ctx, ctx->handoff_off.ptr, off.data(), sizeof(int64_t) * (size_t)B));
    ctx->handoff_off_host = off;
    plan->d_x32 = (float*)ctx->handoff.ptr;
    plan->d_x32_off = (const int64_t*)ctx->handoff_off.ptr;
    plan->h_x32_off = ctx->handoff_off_host.data();
    return F2_OK;
}

static size_t wave_elem(int wave_dtype) { return wave_dtype == F2_WAVE_I16 ? 2 : 8; }

extern "C" {

int f2_erb_filterbank_batch(f2_ctx* ctx, const void* wave, int wave_dtype, const int64_t* offsets,
                            const double* coefs, int B, int C, double* gfb, int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, wave_dtype == F2_WAVE_I16 || wave_dtype == F2_WAVE_F64, F2_ERR_INVALID, "bad wave_dtype %d", wave_dtype);
    F2_TRY(check_batch(ctx, offsets, B, C, mem_space));
    const int64_t total = offsets[B];
    if (B == 0 || C == 0 || total == 0) return F2_OK;
    F2_CHECK(ctx, wave && coefs && gfb, F2_ERR_INVALID, "null data pointer");
    F2_TRY(f2_upload_offsets(ctx, offsets, B));
    F2_TRY(f2_upload_coefs(ctx, coefs, C));
    const void* d_wave = wave;
    double* d_gfb = gfb;
    const size_t out_bytes = sizeof(double) * (size_t)C * (size_t)total;
    const bool staged = mem_space != F2_MEM_DEVICE;
    if (staged) {
        F2_TRY(f2_reserve(ctx, ctx->stage_in, wave_elem(wave_dtype) * (size_t)total));
        F2_TRY(f2_reserve(ctx, ctx->stage_out, out_bytes));
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, wave, wave_elem(wave_dtype) * (size_t)total, hipMemcpyHostToDevice,
                                   ctx->stream));
        d_wave = ctx->stage_in.ptr;
        d_gfb = (double*)ctx->stage_out.ptr;
    }
    F2_TRY(f2_launch_filterbank(ctx, d_wave, wave_dtype, (const int64_t*)ctx->offsets.ptr, offsets,
                                (const double*)ctx->coefs.ptr, B, C, d_gfb));
    if (staged) {
        F2_HIP(ctx, hipMemcpyAsync(gfb, d_gfb, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (mem_space == F2_MEM_HOST) F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return F2_OK;
}

int f2_envelope_batch(f2_ctx* ctx, const double* gfb, const int64_t* offsets, int B, int C, int lpf,
                      double cutoff_hz, int fft_precision, double* env, int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, fft_precision == F2_FFT_F32 || fft_precision == F2_FFT_F64, F2_ERR_INVALID, "bad fft_precision %d", fft_precision);
    F2_CHECK(ctx, !lpf || (cutoff_hz > 0 && cutoff_hz < 8000), F2_ERR_INVALID, "cutoff %g Hz outside (0, 8000)", cutoff_hz);
    F2_TRY(check_batch(ctx, offsets, B, C, mem_space));
    const int64_t total = offsets[B];
    if (B == 0 || C == 0 || total == 0) return F2_OK;
    F2_CHECK(ctx, gfb && env, F2_ERR_INVALID, "null data pointer");
    F2_TRY(f2_upload_offsets(ctx, offsets, B));
    const double* d_gfb = gfb;
    double* d_env = env;
    const size_t bytes = sizeof(double) * (size_t)C * (size_t)total;
    const bool staged = mem_space != F2_MEM_DEVICE;
    if (staged) {
        // separate device buffers for the filterbank rows and the envelopes: rows of 32769..65536 samples then take the
        // on-chip path (which parks intermediate data in the output rows) exactly as they do inside the fused call
        F2_TRY(f2_reserve(ctx, ctx->stage_aux, bytes));
        F2_TRY(f2_reserve(ctx, ctx->stage_out, bytes));
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_aux.ptr, gfb, bytes, hipMemcpyHostToDevice, ctx->stream));
        d_gfb = (const double*)ctx->stage_aux.ptr;
        d_env = (double*)ctx->stage_out.ptr;
    }
    F2_TRY(f2_launch_envelope(ctx, d_gfb, (const int64_t*)ctx->offsets.ptr, offsets, B, C, lpf, cutoff_hz,
                              fft_precision, d_env));
    if (staged) {
        F2_HIP(ctx, hipMemcpyAsync(env, d_env, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (mem_space == F2_MEM_HOST) F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return F2_OK;
}

int f2_filterbank_envelope_fused(f2_ctx* ctx, const void* wave, int wave_dtype, const int64_t* offsets,
                                 const double* coefs, int B, int C, int lpf, double cutoff_hz,
                                 int fft_precision, double* env, double* gfb_or_null, int mem_space) {
    F2_CHECK(nullptr, ctx, F2_ERR_INVALID, "ctx is NULL");
    F2_HIP(ctx, hipSetDevice(ctx->device));
    F2_CHECK(ctx, wave_dtype == F2_WAVE_I16 || wave_dtype == F2_WAVE_F64, F2_ERR_INVALID, "bad wave_dtype %d", wave_dtype);
    F2_CHECK(ctx, fft_precision == F2_FFT_F32 || fft_precision == F2_FFT_F64, F2_ERR_INVALID, "bad fft_precision %d", fft_precision);
    F2_CHECK(ctx, !lpf || (cutoff_hz > 0 && cutoff_hz < 8000), F2_ERR_INVALID, "cutoff %g Hz outside (0, 8000)", cutoff_hz);
    F2_TRY(check_batch(ctx, offsets, B, C, mem_space));
    const int64_t total = offsets[B];
    if (B == 0 || C == 0 || total == 0) return F2_OK;
    F2_CHECK(ctx, wave && coefs && env, F2_ERR_INVALID, "null data pointer");
    F2_TRY(f2_upload_offsets(ctx, offsets, B));
    F2_TRY(f2_upload_coefs(ctx, coefs, C));
    const size_t bytes = sizeof(double) * (size_t)C * (size_t)total;
    const void* d_wave = wave;
    double* d_env = env;
    double* d_gfb = gfb_or_null;
    const bool staged = mem_space != F2_MEM_DEVICE;
    if (staged) {
        F2_TRY(f2_reserve(ctx, ctx->stage_in, wave_elem(wave_dtype) * (size_t)total));
        F2_TRY(f2_reserve(ctx, ctx->stage_out, bytes));
        if (gfb_or_null) F2_TRY(f2_reserve(ctx, ctx->stage_aux, bytes));
        F2_HIP(ctx, hipMemcpyAsync(ctx->stage_in.ptr, wave, wave_elem(wave_dtype) * (size_t)total, hipMemcpyHostToDevice,
                                   ctx->stream));
        d_wave = ctx->stage_in.ptr;
        d_env = (double*)ctx->stage_out.ptr;
        d_gfb = gfb_or_null ? (double*)ctx->stage_aux.ptr : nullptr;
    }
    // Spectral path (f2_spectral.hip): utterances it can serve (float FFT, no GFB output wanted, make_erb_filters-shaped
    // table, 4097..65472 samples with padding to look at) get their envelopes from ONE kernel that never materialises
    // the filterbank rows. Everything else - and any utterance that kernel's accuracy guard flags on the device - goes
    // through the filterbank kernel + envelope kernel below, which skip utterances whose flag is 0.
    const int* d_uflag = nullptr;
    ctx->spec_last_B = 0;
    if (ctx->opt_spectral && fft_precision == F2_FFT_F32 && !d_gfb && ctx->spec_coefs_ok < 0)
        ctx->spec_coefs_ok = f2_spectral_supports_coefs(ctx->coefs_host, C, nullptr, &ctx->spec_min_pad) ? 1 : 0;   // (once per table: ~50 us of logarithms)
    if (ctx->opt_spectral && fft_precision == F2_FFT_F32 && !d_gfb && ctx->spec_coefs_ok == 1) {
        std::vector<int> meta((size_t)B, 1);
        std::vector<int> lists[F2_SPECTRAL_MAX_LOG2H + 1];
        int nspec = 0;
        for (int b = 0; b < B; ++b) {
            const int64_t n = offsets[b + 1] - offsets[b];
            if (!f2_spectral_supports_len(n, ctx->opt_spectral_min_pad >= 0 ? ctx->opt_spectral_min_pad : ctx->spec_min_pad)) continue;
            lists[f2_log2_ceil(n) - 1].push_back(b);
            meta[(size_t)b] = 0;
            ++nspec;
        }
        // a handful of rows cannot hide the serial run of k_tail_state (~0.1 ms for the low channels): small batches
        // (one file of `cnn eval`, cfg1) keep the time-split filterbank kernel + envelope kernel
        if ((int64_t)nspec * C < ctx->opt_spectral_min_rows) nspec = 0;
        if (nspec > 0) {
            size_t pos[F2_SPECTRAL_MAX_LOG2H + 1];
            for (int l = F2_SPECTRAL_MIN_LOG2H; l <= F2_SPECTRAL_MAX_LOG2H; ++l) {
                pos[l] = meta.size();
                meta.insert(meta.end(), lists[l].begin(), lists[l].end());
            }
            if (meta != ctx->spec_meta_host) {   // new batch shape (as f2_upload_offsets: staged, not waited for)
                ctx->spec_meta_host.clear();
                F2_TRY(f2_reserve(ctx, ctx->spec_meta, sizeof(int) * meta.size()));
                F2_TRY(f2_reserve(ctx, ctx->spec_uflag, sizeof(int) * (size_t)B));
                F2_TRY(f2_upload_async(ctx, ctx->spec_meta.ptr, meta.data(), sizeof(int) * meta.size()));
                ctx->spec_meta_host = meta;
            }
            int* uflag = (int*)ctx->spec_uflag.ptr;
            F2_HIP(ctx, hipMemcpyAsync(uflag, ctx->spec_meta.ptr, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, ctx->stream));
            ctx->spec_gdump_rows = 0;
            if (ctx->opt_spectral_guard_dump) {   // diagnostic: rows the spectral kernel does not serve read back as -1
                const size_t gbytes = sizeof(float) * 4 * (size_t)B * (size_t)C;
                F2_TRY(f2_reserve(ctx, ctx->spec_gdump, gbytes));
                F2_HIP(ctx, hipMemsetAsync(ctx->spec_gdump.ptr, 0xff, gbytes, ctx->stream));
                ctx->spec_gdump_rows = (size_t)B * (size_t)C;
            }
            for (int l = F2_SPECTRAL_MIN_LOG2H; l <= F2_SPECTRAL_MAX_LOG2H; ++l) {
                int64_t min_n = INT64_MAX;
                for (int b : lists[l]) min_n = std::min(min_n, offsets[b + 1] - offsets[b]);
                F2_TRY(f2_launch_spectral(ctx, d_wave, wave_dtype, (const int64_t*)ctx->offsets.ptr, (const double*)ctx->coefs.ptr,
                                          C, (const int*)ctx->spec_meta.ptr + pos[l], (int)lists[l].size(), min_n, l, lpf,
                                          cutoff_hz, d_env, uflag));
            }
            d_uflag = uflag;
            ctx->spec_last_B = (size_t)B;
        }
    }
    // Filterbank kernel + envelope kernel queued back to back on the context's stream, no third buffer and no host
    // round trip. When the float64 filterbank output is not wanted and the envelope runs its float32 FFT, the filterbank
    // hands its rows over as float32 inside the ENV buffer itself (half the bytes written and read back;
    // the envelope kernel converts to float32 before its FFT anyway, so the result is bit-identical).
    f2_handoff handoff;
    F2_TRY(f2_plan_handoff(ctx, offsets, B, C, fft_precision, d_gfb != nullptr, &handoff));
    double* k1_out = d_gfb ? d_gfb : d_env;
    F2_TRY(f2_launch_filterbank(ctx, d_wave, wave_dtype, (const int64_t*)ctx->offsets.ptr, offsets,
                                (const double*)ctx->coefs.ptr, B, C, k1_out, &handoff, d_uflag,
                                d_uflag && ctx->spec_meta_host.size() >= (size_t)B ? ctx->spec_meta_host.data() : nullptr));
    F2_TRY(f2_launch_envelope(ctx, k1_out, (const int64_t*)ctx->offsets.ptr, offsets, B, C, lpf, cutoff_hz,
                              fft_precision, d_env, &handoff, d_uflag,
                              d_uflag && ctx->spec_meta_host.size() >= (size_t)B ? ctx->spec_meta_host.data() : nullptr));
    if (staged) {
        F2_HIP(ctx, hipMemcpyAsync(env, d_env, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (gfb_or_null) F2_HIP(ctx, hipMemcpyAsync(gfb_or_null, d_gfb, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (mem_space == F2_MEM_HOST) F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return F2_OK;
}

}  // extern "C"
