// K2 -- Hilbert-magnitude envelope + optional first-order Butterworth low-pass
// (reference: scripts/processing/EnvelopeExtraction.py:20-67 paddedHilbert / lowPassFilter /
// ExtractEnvelopeFromMatrix).
//
// One 256-thread workgroup owns one (utterance, channel) row of n samples, zero-padded to
// M = 2^ceil(log2 n) exactly as the reference does, and keeps the whole transform in LDS:
//
//   1. pack the real row as H = M/2 complex points  z[m] = x[2m] + i x[2m+1]
//   2. Z = FFT_H(z)             radix-8/16 Stockham passes, in place (all reads of a pass are held
//                               in registers across the barrier), twiddles from a float64-built table
//   3. the Hilbert transform h = H[x] is real, so its packed spectrum follows from Z by one pass
//      over the pairs (k, H-k):  W[k] = i sin(t_k) Z[k] + cos(t_k) conj(Z[H-k]),  t_k = 2 pi k / M,
//      W[0] = 0  (derivation in DESIGN.md); scipy.signal.hilbert's analytic signal is x + i h
//   4. w = IFFT_H(W)  (run as conj(FFT(conj W)); only squares of w are used)  ->  h[2m] = Re w[m],
//      h[2m+1] = Im w[m]
//   5. env[n] = sqrt(x[n]^2 + h[n]^2); optional y[n] = b0 (env[n] + env[n-1]) - a1 y[n-1] in float64,
//      time-parallel: each thread runs a contiguous chunk from zero state, the chunk-end values are
//      combined by a multiplicative scan ((-a1)^L per chunk), and the chunk is re-run from its true
//      initial state; the result is staged in LDS so that global stores are fully coalesced.
//
// Two real FFTs of length M thus cost two complex FFTs of length M/2 on 8-byte (f32) points:
// 64 KiB (+pad) of LDS for the 1 s / 16 kHz row of the benchmark, two workgroups per CU.
// Bound: HBM, 16 bytes per sample-channel (8 read + 8 written).
#include <cmath>

#include "f2_internal.h"

namespace {

constexpr int NT = 256;  // threads per workgroup

template <typename F>
struct cpx {
    F re, im;
};
template <typename F>
__device__ __forceinline__ cpx<F> operator+(cpx<F> a, cpx<F> b) { return {a.re + b.re, a.im + b.im}; }
template <typename F>
__device__ __forceinline__ cpx<F> operator-(cpx<F> a, cpx<F> b) { return {a.re - b.re, a.im - b.im}; }
template <typename F>
__device__ __forceinline__ cpx<F> cmul(cpx<F> a, cpx<F> w) {
    return {a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re};
}

// cos/sin(2 pi j / 32), j = 0..15
__device__ constexpr double kCos32[16] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                                          0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                                          0.19509032201612826785, 0.0, -0.19509032201612826785, -0.38268343236508977173,
                                          -0.55557023301960222474, -0.70710678118654752440, -0.83146961230254523708,
                                          -0.92387953251128675613, -0.98078528040323044913};
__device__ constexpr double kSin32[16] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                                          0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                                          0.98078528040323044913, 1.0, 0.98078528040323044913, 0.92387953251128675613,
                                          0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                                          0.38268343236508977173, 0.19509032201612826785};

// a * exp(-2 pi i K / R)
template <int R, int K, typename F>
__device__ __forceinline__ cpx<F> mulw(cpx<F> a) {
    if constexpr (K == 0) {
        return a;
    } else if constexpr (4 * K == R) {
        return {a.im, -a.re};
    } else if constexpr (8 * K == R) {
        const F h = F(0.70710678118654752440);
        return {(a.re + a.im) * h, (a.im - a.re) * h};
    } else if constexpr (8 * K == 3 * R) {
        const F h = F(0.70710678118654752440);
        return {(a.im - a.re) * h, -(a.re + a.im) * h};
    } else {
        const F c = F(kCos32[K * 32 / R]), s = F(kSin32[K * 32 / R]);
        return {a.re * c + a.im * s, a.im * c - a.re * s};
    }
}

template <int R, int K, typename F>
__device__ __forceinline__ void combine(cpx<F>* v, const cpx<F>* e, const cpx<F>* o) {
    if constexpr (K < R / 2) {
        const cpx<F> t = mulw<R, K>(o[K]);
        v[K] = e[K] + t;
        v[K + R / 2] = e[K] - t;
        combine<R, K + 1>(v, e, o);
    }
}

// forward DFT of R points held in registers, natural order in and out
template <int R, typename F>
__device__ __forceinline__ void dft(cpx<F>* v) {
    if constexpr (R == 2) {
        const cpx<F> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    } else if constexpr (R > 2) {
        cpx<F> e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        dft<R / 2>(e);
        dft<R / 2>(o);
        combine<R, 0>(v, e, o);
    }
}

// pass plan: ceil(LOG2H/4) passes, bits spread evenly (13 -> 4,3,3,3; 12 -> 4,4,4)
constexpr int plan_npass(int log2h) { return (log2h + 3) / 4; }
constexpr int plan_bits(int log2h, int pass) {
    const int np = plan_npass(log2h);
    return np == 0 ? 0 : log2h / np + (pass < log2h % np ? 1 : 0);
}
constexpr int plan_shift(int log2h, int pass) {  // log2 of the stride entering `pass`
    int s = 0;
    for (int i = 0; i < pass; ++i) s += plan_bits(log2h, i);
    return s;
}

// LDS index padding of the complex array: one extra slot per 16 (keeps the stride-R writes of the
// early passes off a single bank)
__device__ __forceinline__ int cpad(int i) { return i + (i >> 4); }
constexpr int cpad_size(int h) { return h + (h >> 4) + 1; }

// exp(-2 pi i t / H) from the half-circle table V[k] = exp(-2 pi i k / M), k < H
template <typename F, int LOG2H>
__device__ __forceinline__ cpx<F> twiddle_h(const cpx<F>* __restrict__ V, int t) {
    constexpr int H = 1 << LOG2H;
    const int u = 2 * t;
    if (u < H) return V[u];
    const cpx<F> w = V[u - H];
    return {-w.re, -w.im};
}

template <typename F, int LOG2H, int PASS>
__device__ __forceinline__ void fft_pass(cpx<F>* lds, const cpx<F>* __restrict__ V, int tid) {
    constexpr int H = 1 << LOG2H;
    constexpr int LOG2R = plan_bits(LOG2H, PASS);
    constexpr int R = 1 << LOG2R;
    constexpr int S = 1 << plan_shift(LOG2H, PASS);
    constexpr int NB = H / R;
    constexpr int ITER = (NB + NT - 1) / NT;
    constexpr bool LAST = (S * R == H);
    cpx<F> v[ITER][R];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int bf = tid + i * NT;
        if (bf < NB) {
#pragma unroll
            for (int j = 0; j < R; ++j) v[i][j] = lds[cpad(bf + j * NB)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int bf = tid + i * NT;
        if (bf < NB) {
            dft<R>(v[i]);
            const int q = bf & (S - 1);
            const int ps = bf - q;  // p * S
            const int base = q + ps * R;
            lds[cpad(base)] = v[i][0];
#pragma unroll
            for (int k = 1; k < R; ++k) {
                cpx<F> o = v[i][k];
                if constexpr (!LAST) o = cmul(o, twiddle_h<F, LOG2H>(V, ps * k));
                lds[cpad(base + S * k)] = o;
            }
        }
    }
    __syncthreads();
}

template <typename F, int LOG2H, int PASS = 0>
__device__ __forceinline__ void fft_all(cpx<F>* lds, const cpx<F>* __restrict__ V, int tid) {
    if constexpr (PASS < plan_npass(LOG2H)) {
        fft_pass<F, LOG2H, PASS>(lds, V, tid);
        fft_all<F, LOG2H, PASS + 1>(lds, V, tid);
    }
}

__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, d);
    hi = __shfl_up(hi, d);
    return __hiloint2double(hi, lo);
}

struct EnvParams {
    const double* gfb;
    double* env;
    const int64_t* offsets;
    const int* ulist;  // utterances served by this launch (NULL: identity)
    int C;
    int lpf;
    double b0, a1;     // y[n] = b0 (e[n] + e[n-1]) - a1 y[n-1]
};

template <typename F, int LOG2H>
__global__ __launch_bounds__(NT) void k_envelope(EnvParams P, const cpx<F>* __restrict__ V) {
    constexpr int H = 1 << LOG2H;
    constexpr int M = 2 * H;
    constexpr int CS = cpad_size(H);
    constexpr int L = M >= NT ? M / NT : 1;   // samples per thread in the low-pass (contiguous chunk)
    constexpr int TP = NT + 1;                // row pitch of the transposed envelope image
    constexpr int RS = L * TP;
    constexpr int LDS_BYTES = (CS * 2 > RS ? CS * 2 : RS) * (int)sizeof(F);
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    __shared__ double wave_tot[NT / 64];
    cpx<F>* lds = reinterpret_cast<cpx<F>*>(smem);
    F* rl = reinterpret_cast<F*>(smem);

    const int tid = threadIdx.x;
    const int u = blockIdx.x / P.C;
    const int c = blockIdx.x - u * P.C;
    const int b = P.ulist ? P.ulist[u] : u;
    const int64_t off = P.offsets[b];
    const int n = (int)(P.offsets[b + 1] - off);
    const size_t row = (size_t)P.C * (size_t)off + (size_t)c * (size_t)n;
    const double* __restrict__ x = P.gfb + row;
    double* __restrict__ y = P.env + row;

    // 1. pack
    for (int m = tid; m < H; m += NT) {
        const int i0 = 2 * m;
        const F a = i0 < n ? (F)x[i0] : F(0);
        const F bb = i0 + 1 < n ? (F)x[i0 + 1] : F(0);
        lds[cpad(m)] = {a, bb};
    }
    __syncthreads();
    // 2. forward transform
    fft_all<F, LOG2H>(lds, V, tid);
    // 3. packed spectrum of the Hilbert transform, conjugated and scaled by 1/H for step 4
    {
        const F sc = F(1.0 / H);
        for (int k = tid; k <= H / 2; k += NT) {
            if (k == 0) {
                lds[cpad(0)] = {F(0), F(0)};
                continue;
            }
            const cpx<F> vk = V[k];  // (cos t, -sin t)
            const F cs = vk.re * sc, sn = -vk.im * sc;
            const cpx<F> zk = lds[cpad(k)];
            if (2 * k == H) {
                // t = pi/2: W = i Z  ->  conj(W) = -i conj(Z) = (-zim, -zre)
                lds[cpad(k)] = {-zk.im * sc, -zk.re * sc};
            } else {
                const cpx<F> zh = lds[cpad(H - k)];
                // W[k]   = i sn Z[k]   + cs conj(Z[H-k]);  W[H-k] = i sn Z[H-k] - cs conj(Z[k])
                const cpx<F> wk = {-sn * zk.im + cs * zh.re, sn * zk.re - cs * zh.im};
                const cpx<F> wh = {-sn * zh.im - cs * zk.re, sn * zh.re + cs * zk.im};
                lds[cpad(k)] = {wk.re, -wk.im};
                lds[cpad(H - k)] = {wh.re, -wh.im};
            }
        }
    }
    __syncthreads();
    // 4. inverse transform (forward transform of the conjugate)
    fft_all<F, LOG2H>(lds, V, tid);

    // 5. magnitude (in F: the FFT already limits the accuracy to F)
    constexpr int PT = (H + NT - 1) / NT;  // packed points per thread
    if (!P.lpf) {
#pragma unroll 4
        for (int m = tid; m < H; m += NT) {
            const cpx<F> w = lds[cpad(m)];
            const int i0 = 2 * m;
            if (i0 < n) {
                const F a = (F)x[i0];
                y[i0] = (double)sqrt(a * a + w.re * w.re);
            }
            if (i0 + 1 < n) {
                const F a = (F)x[i0 + 1];
                y[i0 + 1] = (double)sqrt(a * a + w.im * w.im);
            }
        }
        return;
    }
    // With the low-pass the envelope goes back to LDS, TRANSPOSED: thread t will own the contiguous
    // samples [t*L, (t+1)*L), so sample n = t*L + j is kept at j*TP + t (TP = 257: conflict-free for
    // the per-chunk sweeps and for the coalesced copy-out, 2-way on this scatter).
    auto tpos = [](int i) { return (i % L) * TP + i / L; };
    F e0[PT], e1[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int m = tid + j * NT;
        e0[j] = e1[j] = F(0);
        if (m < H) {
            const cpx<F> w = lds[cpad(m)];
            const int i0 = 2 * m;
            const F a = i0 < n ? (F)x[i0] : F(0);
            const F bb = i0 + 1 < n ? (F)x[i0 + 1] : F(0);
            e0[j] = sqrt(a * a + w.re * w.re);
            e1[j] = sqrt(bb * bb + w.im * w.im);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int m = tid + j * NT;
        if (m < H) {
            rl[tpos(2 * m)] = e0[j];
            rl[tpos(2 * m + 1)] = e1[j];
        }
    }
    __syncthreads();

    // low-pass: chunked recurrence + multiplicative scan over the 256 chunks
    const int n0 = tid * L;
    const double b0 = P.b0, na1 = -P.a1;
    const double eprev = (n0 > 0 && n0 <= n) ? (double)rl[tpos(n0 - 1)] : 0.0;
    double yz = 0.0;
    {
        double ep = eprev;
#pragma unroll 8
        for (int j = 0; j < L; ++j) {
            const double e = (double)rl[j * TP + tid];   // samples past n hold |0| = 0
            yz = fma(na1, yz, b0 * (e + ep));
            ep = e;
        }
    }
    // g = (-a1)^L
    double g = na1;
#pragma unroll
    for (int s = 1; s < L; s <<= 1) g *= g;
    // inclusive scan inside the wave: v_t = sum_{j<=t} g^(t-j) yz_j
    const int lane = tid & 63, wv = tid >> 6;
    double v = yz, gd = g;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = shfl_up_f64(v, d);
        if (lane >= d) v = fma(gd, up, v);
        gd *= gd;
    }
    // gd == g^64 now
    if (lane == 63) wave_tot[wv] = v;
    __syncthreads();
    double carry = 0.0;  // true y at the end of the previous wave's last chunk
    for (int w2 = 0; w2 < wv; ++w2) carry = fma(gd, carry, wave_tot[w2]);
    // g^(lane+1)
    double gl = 1.0, gp = g;
    for (int bits = lane + 1; bits; bits >>= 1) {
        if (bits & 1) gl *= gp;
        gp *= gp;
    }
    const double incl = fma(gl, carry, v);                  // true y at the end of this chunk
    double yprev = shfl_up_f64(incl, 1);                    // ... of the previous chunk
    if (lane == 0) yprev = carry;
    {
        double ep = eprev, yy = yprev;
#pragma unroll 8
        for (int j = 0; j < L; ++j) {
            const double e = (double)rl[j * TP + tid];
            yy = fma(na1, yy, b0 * (e + ep));
            ep = e;
            rl[j * TP + tid] = (F)yy;
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = tid; i < n; i += NT) y[i] = (double)rl[tpos(i)];
}

template <typename F>
using EnvKernel = void (*)(EnvParams, const cpx<F>*);

template <typename F, int... L>
constexpr EnvKernel<F> kernel_table_impl(int log2h, std::integer_sequence<int, L...>) {
    EnvKernel<F> tab[] = {k_envelope<F, L>...};
    return tab[log2h];
}

template <typename F, int MAXL>
EnvKernel<F> kernel_for(int log2h) {
    return kernel_table_impl<F>(log2h, std::make_integer_sequence<int, MAXL + 1>{});
}

constexpr int MAX_LOG2H_F32 = 14;  // rows up to 32768 samples
constexpr int MAX_LOG2H_F64 = 13;  // rows up to 16384 samples

template <typename F>
int ensure_twiddles(f2_ctx* ctx, int log2h, f2_scratch& slot) {
    if (slot.ptr) return F2_OK;
    const int H = 1 << log2h;
    std::vector<cpx<F>> host((size_t)H);
    const long double step = 2.0L * 3.14159265358979323846264338327950288L / (long double)(2 * H);
    for (int k = 0; k < H; ++k) {
        host[k].re = (F)cosl(step * k);
        host[k].im = (F)(-sinl(step * k));
    }
    F2_TRY(f2_reserve(ctx, slot, sizeof(cpx<F>) * (size_t)H));
    F2_HIP(ctx, hipMemcpyAsync(slot.ptr, host.data(), sizeof(cpx<F>) * (size_t)H, hipMemcpyHostToDevice, ctx->stream));
    F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return F2_OK;
}

}  // namespace

int f2_launch_envelope(f2_ctx* ctx, const double* d_gfb, const int64_t* d_offsets, const int64_t* h_offsets,
                       int B, int C, int lpf, double cutoff_hz, int precision, double* d_env) {
    EnvParams P;
    P.gfb = d_gfb;
    P.env = d_env;
    P.offsets = d_offsets;
    P.C = C;
    P.lpf = lpf ? 1 : 0;
    // butter(1, cutoff/8000): bilinear transform of 1/(s+1), pre-warped (EnvelopeExtraction.py:47)
    const double k = lpf ? tan(3.14159265358979323846 * cutoff_hz / 16000.0) : 0.0;
    P.b0 = k / (1.0 + k);
    P.a1 = (k - 1.0) / (k + 1.0);

    // group the utterances by padded length (one kernel instantiation per FFT size)
    std::vector<std::vector<int>> groups(32);
    for (int b = 0; b < B; ++b) {
        const int64_t n = h_offsets[b + 1] - h_offsets[b];
        if (n <= 0) continue;
        const int log2m = n <= 2 ? 1 : f2_log2_ceil(n);
        const int log2h = log2m - 1;
        const int maxl = precision == F2_FFT_F32 ? MAX_LOG2H_F32 : MAX_LOG2H_F64;
        F2_CHECK(ctx, log2h <= maxl, F2_ERR_UNSUPPORTED,
                 "utterance %d has %lld samples; rows longer than %d samples are not supported with this FFT precision",
                 b, (long long)n, 2 << maxl);
        groups[log2h].push_back(b);
    }
    size_t list_elems = 0;
    int ngroups = 0;
    for (auto& g : groups)
        if (!g.empty()) {
            ++ngroups;
            list_elems += g.size();
        }
    const bool identity = ngroups == 1 && (int)list_elems == B;
    int* d_lists = nullptr;
    if (!identity && ngroups > 0) {
        std::vector<int> flat;
        flat.reserve(list_elems);
        for (auto& g : groups) flat.insert(flat.end(), g.begin(), g.end());
        F2_TRY(f2_reserve(ctx, ctx->work2, sizeof(int) * flat.size()));
        F2_HIP(ctx, hipMemcpyAsync(ctx->work2.ptr, flat.data(), sizeof(int) * flat.size(), hipMemcpyHostToDevice,
                                   ctx->stream));
        F2_HIP(ctx, hipStreamSynchronize(ctx->stream));
        d_lists = (int*)ctx->work2.ptr;
    }
    size_t pos = 0;
    for (int log2h = 0; log2h < 32; ++log2h) {
        const auto& g = groups[log2h];
        if (g.empty()) continue;
        P.ulist = identity ? nullptr : d_lists + pos;
        pos += g.size();
        const dim3 grid((unsigned)(g.size() * (size_t)C)), block(NT);
        F2_TRY(f2_prof_begin(ctx, F2_K_ENVELOPE));
        if (precision == F2_FFT_F32) {
            F2_TRY(ensure_twiddles<float>(ctx, log2h, ctx->tw[0][log2h]));
            const EnvKernel<float> kern = kernel_for<float, MAX_LOG2H_F32>(log2h);
            hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, P, (const cpx<float>*)ctx->tw[0][log2h].ptr);
        } else {
            F2_TRY(ensure_twiddles<double>(ctx, log2h, ctx->tw[1][log2h]));
            const EnvKernel<double> kern = kernel_for<double, MAX_LOG2H_F64>(log2h);
            hipLaunchKernelGGL(kern, grid, block, 0, ctx->stream, P, (const cpx<double>*)ctx->tw[1][log2h].ptr);
        }
        F2_HIP(ctx, hipGetLastError());
        F2_TRY(f2_prof_end(ctx, F2_K_ENVELOPE));
    }
    return F2_OK;
}
