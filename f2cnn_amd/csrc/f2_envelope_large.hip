// K2, long rows: the same Hilbert-magnitude (+ low-pass) envelope as f2_envelope.hip for rows whose packed
// transform (H = M/2 complex points) does not fit in LDS, i.e. utterances longer than ~2 s at 16 kHz (most
// TIMIT sentences). Same mathematics (reference: scripts/processing/EnvelopeExtraction.py:20-67), but the
// H-point Stockham transform runs as radix-16 passes over ping-pong buffers in global memory (L2 / Infinity
// Cache resident for typical sizes), one launch per pass for all C rows of an utterance:
//
//   pack -> passes (16,16,..,r) -> Hilbert pair sweep -> passes -> magnitude [-> low-pass, one workgroup/row]
//
// Throughput is secondary here (about 40 bytes of global traffic per sample-channel and pass pair); the LDS
// kernel stays the fast path for rows up to 32768 samples.
#include <cmath>

#include "f2_envelope_core.h"

using namespace f2fft;

namespace {

constexpr int LT = 256;

template <typename F>
__global__ __launch_bounds__(LT) void k_pack(const double* __restrict__ x, int n, int H, int C, cpx<F>* __restrict__ a) {
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    if (idx >= (int64_t)C * H) return;
    const int row = (int)(idx / H), m = (int)(idx - (int64_t)row * H);
    const double* xr = x + (size_t)row * n;
    const int i0 = 2 * m;
    a[idx] = {i0 < n ? (F)xr[i0] : F(0), i0 + 1 < n ? (F)xr[i0 + 1] : F(0)};
}

// one Stockham pass of radix R over every row: thread = butterfly bf, inputs src[bf + j*H/R],
// outputs dst[q + S*(R*p + k)] * exp(-2 pi i p S k / H), p = bf / S, q = bf % S
template <typename F, int R>
__global__ __launch_bounds__(LT) void k_pass(const cpx<F>* __restrict__ src, cpx<F>* __restrict__ dst,
                                             const cpx<F>* __restrict__ W, int H, int S, int C) {
    const int NB = H / R;
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    if (idx >= (int64_t)C * NB) return;
    const int row = (int)(idx / NB), bf = (int)(idx - (int64_t)row * NB);
    const cpx<F>* s = src + (size_t)row * H;
    cpx<F>* d = dst + (size_t)row * H;
    cpx<F> v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = s[bf + j * NB];
    dft<R>(v);
    const int q = bf % S, ps = bf - q;
    const int base = q + ps * R;
    const bool last = (int64_t)S * R == H;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        cpx<F> o = v[brev<R>(k)];
        if (k > 0 && !last) o = cmul(o, W[(int64_t)ps * k]);
        d[base + S * k] = o;
    }
}

// W[k] = i sin(t_k) Z[k] + cos(t_k) conj(Z[H-k]), W[0] = 0; stored conjugated and scaled by 1/H, in place
template <typename F>
__global__ __launch_bounds__(LT) void k_pairs(cpx<F>* __restrict__ a, const cpx<F>* __restrict__ V, int H, int C) {
    const int half = H / 2;
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    if (idx >= (int64_t)C * (half + 1)) return;
    const int row = (int)(idx / (half + 1)), k = (int)(idx - (int64_t)row * (half + 1));
    cpx<F>* z = a + (size_t)row * H;
    const F sc = F(1.0 / H);
    if (k == 0) {
        z[0] = {F(0), F(0)};
        return;
    }
    const cpx<F> zk = z[k];
    if (k == half) {
        z[k] = {-zk.im * sc, -zk.re * sc};
        return;
    }
    const cpx<F> vk = V[k];
    const F cs = vk.re * sc, sn = -vk.im * sc;
    const cpx<F> zh = z[H - k];
    z[k] = {-sn * zk.im + cs * zh.re, -(sn * zk.re - cs * zh.im)};
    z[H - k] = {-sn * zh.im - cs * zk.re, -(sn * zh.re + cs * zk.im)};
}

// env[n] = sqrt(x^2 + h^2), h[2m] = Re w[m], h[2m+1] = Im w[m]; to the float64 output (no low-pass) or to the
// float scratch e (low-pass follows)
template <typename F>
__global__ __launch_bounds__(LT) void k_magnitude(const double* __restrict__ x, const cpx<F>* __restrict__ w, int n,
                                                  int H, int C, double* __restrict__ y, F* __restrict__ e) {
    const int64_t idx = (int64_t)blockIdx.x * LT + threadIdx.x;
    if (idx >= (int64_t)C * H) return;
    const int row = (int)(idx / H), m = (int)(idx - (int64_t)row * H);
    const cpx<F> wm = w[idx];
    const int i0 = 2 * m;
    const size_t r = (size_t)row * n;
    if (i0 < n) {
        const F a = (F)x[r + i0];
        const F v = fsqrt(a * a + wm.re * wm.re);
        if (e) e[r + i0] = v; else y[r + i0] = (double)v;
    }
    if (i0 + 1 < n) {
        const F a = (F)x[r + i0 + 1];
        const F v = fsqrt(a * a + wm.im * wm.im);
        if (e) e[r + i0 + 1] = v; else y[r + i0 + 1] = (double)v;
    }
}

// y[n] = b0 (e[n] + e[n-1]) - a1 y[n-1] from zero state; one 1024-thread workgroup per row, contiguous chunk
// per thread: zero-state run, multiplicative scan of the chunk ends, second run from the true state
constexpr int PT_ = 1024;
template <typename F>
__global__ __launch_bounds__(PT_) void k_lowpass(const F* __restrict__ e, int n, double b0, double a1,
                                                 double* __restrict__ y) {
    __shared__ double wave_tot[PT_ / 64];
    const int tid = threadIdx.x;
    const size_t r = (size_t)blockIdx.x * n;
    const int L = (n + PT_ - 1) / PT_;
    const int n0 = tid * L, n1 = min(n0 + L, n);
    const double na1 = -a1;
    const double eprev = (n0 > 0 && n0 <= n) ? (double)e[r + n0 - 1] : 0.0;
    double yz = 0.0, ep = eprev;
    for (int i = n0; i < n0 + L; ++i) {
        if (i < n1) {
            const double ev = (double)e[r + i];
            yz = fma(na1, yz, b0 * (ev + ep));
            ep = ev;
        } else {
            yz *= na1;   // uniform chunk multiplier (-a1)^L
        }
    }
    double g = 1.0, gp = na1;
    for (int bits = L; bits; bits >>= 1) {
        if (bits & 1) g *= gp;
        gp *= gp;
    }
    const int lane = tid & 63, wv = tid >> 6;
    double sc = yz, gd = g;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = shfl_up_f64(sc, d);
        if (lane >= d) sc = fma(gd, up, sc);
        gd *= gd;
    }
    if (lane == 63) wave_tot[wv] = sc;
    __syncthreads();
    double carry = 0.0;
    for (int w2 = 0; w2 < wv; ++w2) carry = fma(gd, carry, wave_tot[w2]);
    double gl = 1.0;
    gp = g;
    for (int bits = lane + 1; bits; bits >>= 1) {
        if (bits & 1) gl *= gp;
        gp *= gp;
    }
    const double incl = fma(gl, carry, sc);
    double yy = shfl_up_f64(incl, 1);
    if (lane == 0) yy = carry;
    ep = eprev;
    for (int i = n0; i < n1; ++i) {
        const double ev = (double)e[r + i];
        yy = fma(na1, yy, b0 * (ev + ep));
        ep = ev;
        y[r + i] = yy;
    }
}

template <typename F>
int ensure_large_tables(f2_ctx* ctx, int log2h, f2_scratch& slot) {
    if (slot.ptr) return F2_OK;
    const int64_t H = int64_t(1) << log2h;
    const long double tau = 2.0L * 3.14159265358979323846264338327950288L;
    std::vector<cpx<F>> host((size_t)H + (size_t)H / 2 + 1);
    for (int64_t t = 0; t < H; ++t) {   // W[t] = exp(-2 pi i t / H)
        const long double ang = tau * (long double)t / (long double)H;
        host[(size_t)t] = {(F)cosl(ang), (F)(-sinl(ang))};
    }
    for (int64_t k = 0; k <= H / 2; ++k) {   // V[k] = exp(-2 pi i k / M)
        const long double ang = tau * (long double)k / (long double)(2 * H);
        host[(size_t)(H + k)] = {(F)cosl(ang), (F)(-sinl(ang))};
    }
    F2_TRY(f2_reserve(ctx, slot, sizeof(cpx<F>) * host.size()));
    F2_TRY(f2_upload_async(ctx, slot.ptr, host.data(), sizeof(cpx<F>) * host.size()));
    return F2_OK;
}

template <typename F>
int fft_passes(f2_ctx* ctx, cpx<F>*& cur, cpx<F>*& other, const cpx<F>* W, int log2h, int C) {
    const int H = 1 << log2h;
    int shift = 0;
    while (shift < log2h) {
        const int bits = log2h - shift >= 4 ? 4 : log2h - shift;
        const int R = 1 << bits, S = 1 << shift;
        const int64_t threads = (int64_t)C * (H / R);
        const dim3 grid((unsigned)((threads + LT - 1) / LT)), block(LT);
        switch (bits) {
            case 4: hipLaunchKernelGGL((k_pass<F, 16>), grid, block, 0, ctx->stream, cur, other, W, H, S, C); break;
            case 3: hipLaunchKernelGGL((k_pass<F, 8>), grid, block, 0, ctx->stream, cur, other, W, H, S, C); break;
            case 2: hipLaunchKernelGGL((k_pass<F, 4>), grid, block, 0, ctx->stream, cur, other, W, H, S, C); break;
            default: hipLaunchKernelGGL((k_pass<F, 2>), grid, block, 0, ctx->stream, cur, other, W, H, S, C); break;
        }
        F2_HIP(ctx, hipGetLastError());
        std::swap(cur, other);
        shift += bits;
    }
    return F2_OK;
}

template <typename F>
int run_large(f2_ctx* ctx, const double* d_x, double* d_y, int64_t n, int C, int lpf, double b0, double a1, int prec) {
    const int log2m = f2_log2_ceil(n), log2h = log2m - 1;
    const int H = 1 << log2h;
    F2_TRY(ensure_large_tables<F>(ctx, log2h, ctx->tw_large[prec][log2h]));
    const cpx<F>* W = (const cpx<F>*)ctx->tw_large[prec][log2h].ptr;
    const cpx<F>* V = W + H;
    const size_t cbytes = sizeof(cpx<F>) * (size_t)C * H;
    const size_t ebytes = lpf ? sizeof(F) * (size_t)C * (size_t)n : 0;
    F2_TRY(f2_reserve(ctx, ctx->work, 2 * cbytes + ebytes + 256));
    cpx<F>* bufa = (cpx<F>*)ctx->work.ptr;
    cpx<F>* bufb = bufa + (size_t)C * H;
    F* e = lpf ? (F*)(bufb + (size_t)C * H) : nullptr;
    const dim3 block(LT);
    const dim3 gridH((unsigned)(((int64_t)C * H + LT - 1) / LT));
    hipLaunchKernelGGL(k_pack<F>, gridH, block, 0, ctx->stream, d_x, (int)n, H, C, bufa);
    F2_HIP(ctx, hipGetLastError());
    cpx<F>*cur = bufa, *other = bufb;
    F2_TRY(fft_passes<F>(ctx, cur, other, W, log2h, C));
    const dim3 gridP((unsigned)(((int64_t)C * (H / 2 + 1) + LT - 1) / LT));
    hipLaunchKernelGGL(k_pairs<F>, gridP, block, 0, ctx->stream, cur, V, H, C);
    F2_HIP(ctx, hipGetLastError());
    F2_TRY(fft_passes<F>(ctx, cur, other, W, log2h, C));
    hipLaunchKernelGGL(k_magnitude<F>, gridH, block, 0, ctx->stream, d_x, cur, (int)n, H, C, d_y, e);
    F2_HIP(ctx, hipGetLastError());
    if (lpf) {
        hipLaunchKernelGGL(k_lowpass<F>, dim3((unsigned)C), dim3(PT_), 0, ctx->stream, e, (int)n, b0, a1, d_y);
        F2_HIP(ctx, hipGetLastError());
    }
    return F2_OK;
}

}  // namespace

// One utterance (all C rows, n samples each, contiguous (C,n) block) through the global-memory transform.
int f2_launch_envelope_large(f2_ctx* ctx, const double* d_x, double* d_y, int64_t n, int C, int lpf, double b0,
                             double a1, int precision) {
    F2_CHECK(ctx, n > 2 && f2_log2_ceil(n) <= F2_MAX_LOG2M_LARGE, F2_ERR_UNSUPPORTED,
             "rows of %lld samples are not supported (limit 2^%d)", (long long)n, F2_MAX_LOG2M_LARGE);
    if (precision == F2_FFT_F32) return run_large<float>(ctx, d_x, d_y, n, C, lpf, b0, a1, 0);
    return run_large<double>(ctx, d_x, d_y, n, C, lpf, b0, a1, 1);
}
