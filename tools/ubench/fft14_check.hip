// Check of the merged passes 1 + 2 of the 16384-point transform (fft14_pass12_merged, f2_fft13_merged.h) against the four
// separate passes of f2_fft_lds.h on one random row.
#include <cstdio>
#include <vector>
#include <random>
#include "../../f2cnn_amd/csrc/f2_fft_lds.h"
using namespace f2fft;
namespace {
#include "../../f2cnn_amd/csrc/f2_fft13_merged.h"
template <bool MERGED>
__global__ __launch_bounds__(1024) void k(const cpx<float>* in, cpx<float>* out, const cpx<float>* tw) {
    constexpr int LOG2H = 14, NT = 1024, PT = 16, H = 16384;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cpx<float>* lds = reinterpret_cast<cpx<float>*>(smem);
    constexpr int TWL = plan_tw_lds_count(LOG2H);
    cpx<float>* twl = lds + cpad_size(H);
    const int tid = threadIdx.x;
    for (int i = tid; i < TWL; i += NT) twl[i] = tw[plan_tw_offset(LOG2H, 1) + i];
    cpx<float> v[PT];
    for (int j = 0; j < 16; ++j) v[j] = in[tid + 1024 * j];
    fft_pass<float, LOG2H, 0, true, false, PT, NT, false, true>(lds, tw, twl, tid, v);
    if (MERGED) {
        fft14_pass12_merged<NT, PT>(lds, twl, tid, v);
    } else {
        fft_pass<float, LOG2H, 1, false, false, PT, NT, false, true>(lds, tw, twl, tid, v);
        fft_pass<float, LOG2H, 2, false, false, PT, NT, false, true>(lds, tw, twl, tid, v);
    }
    fft_pass<float, LOG2H, 3, false, true, PT, NT, false, true>(lds, tw, twl, tid, v);
    for (int j = 0; j < 16; ++j) out[tid + 1024 * j] = v[brev<16>(j)];
}
}  // namespace
int main() {
    const int H = 16384, log2h = 14;
    std::vector<cpx<float>> host;
    const long double tau = 2.0L * 3.14159265358979323846264338327950288L;
    for (int pass = 0; pass < plan_npass(log2h); ++pass) {
        const int R = 1 << plan_bits(log2h, pass), S = 1 << plan_shift(log2h, pass);
        if (S * R == H) continue;
        const int np = H / R / S;
        for (int kk = 1; kk < R; ++kk)
            for (int p = 0; p < np; ++p) {
                const long double ang = tau * (long double)((long long)p * S * kk % H) / (long double)H;
                host.push_back({(float)cosl(ang), (float)(-sinl(ang))});
            }
    }
    std::vector<cpx<float>> x(H), a(H), b(H);
    std::mt19937 g(1);
    std::normal_distribution<float> nd;
    for (auto& e : x) e = {nd(g), nd(g)};
    cpx<float>*dx, *da, *db, *dt;
    (void)hipMalloc(&dx, H * 8); (void)hipMalloc(&da, H * 8); (void)hipMalloc(&db, H * 8); (void)hipMalloc(&dt, host.size() * 8);
    (void)hipMemcpy(dx, x.data(), H * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dt, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (cpad_size(H) + plan_tw_lds_count(log2h)) * 8;
    (void)hipFuncSetAttribute((const void*)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<false><<<1, 1024, lds>>>(dx, da, dt);
    k<true><<<1, 1024, lds>>>(dx, db, dt);
    (void)hipMemcpy(a.data(), da, H * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), db, H * 8, hipMemcpyDeviceToHost);
    double md = 0, mx = 0;
    int bad = 0;
    for (int i = 0; i < H; ++i) {
        const double d = std::hypot((double)a[i].re - b[i].re, (double)a[i].im - b[i].im);
        md = std::max(md, d);
        mx = std::max(mx, std::hypot((double)a[i].re, (double)a[i].im));
        if (d > 1e-3 && bad++ < 5) printf("  mismatch at %d: %g %g vs %g %g\n", i, a[i].re, a[i].im, b[i].re, b[i].im);
    }
    printf("merged vs separate passes: max |diff| %.3e (max |X| %.1f), %s\n", md, mx, hipGetErrorString(hipGetLastError()));
    return md < 1e-3 ? 0 : 1;
}
