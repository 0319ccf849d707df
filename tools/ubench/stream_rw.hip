// Micro-benchmark: what HBM gives a kernel with K2's traffic shape - every sample-channel read as 4 bytes (float) and
// written as 8 bytes (double), rows of 16000 samples, one workgroup per row - and the same for a pure read and a
// pure write. Also reports the core clock under that load (s_memtime ticks at the shader clock, s_memrealtime at
// 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void k_rw(const float* __restrict__ in, double* __restrict__ out, int n, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const size_t row = (size_t)blockIdx.x * n;
    const float2* x = reinterpret_cast<const float2*>(in + row);
    double2* y = reinterpret_cast<double2*>(out + row);
    float2 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = threadIdx.x + 512 * j;
        v[j] = 2 * m < n ? x[m] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = threadIdx.x + 512 * j;
        if (2 * m < n) y[m] = make_double2((double)v[j].x * 1.5, (double)v[j].y * 1.5);
    }
    if (threadIdx.x == 0 && blockIdx.x == 1000) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
__global__ __launch_bounds__(512) void k_r(const float* __restrict__ in, float* __restrict__ sink, int n) {
    const float2* x = reinterpret_cast<const float2*>(in + (size_t)blockIdx.x * n);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = threadIdx.x + 512 * j;
        if (2 * m < n) { const float2 t = x[m]; acc += t.x + t.y; }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(512) void k_w(double* __restrict__ out, int n) {
    double2* y = reinterpret_cast<double2*>(out + (size_t)blockIdx.x * n);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int m = threadIdx.x + 512 * j;
        if (2 * m < n) y[m] = make_double2(1.0 * m, 2.0);
    }
}
int main() {
    const int rows = 128000, n = 16000;
    float* in; double* out; unsigned long long* clk;
    hipMalloc(&in, sizeof(float) * (size_t)rows * n);
    hipMalloc(&out, sizeof(double) * (size_t)rows * n);
    hipMalloc(&clk, 16);
    hipMemset(in, 0, sizeof(float) * (size_t)rows * n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double bytes, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-22s %.3f ms  %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    const double samples = (double)rows * n;
    timeit("read 4 B + write 8 B", samples * 12, [&] { k_rw<<<rows, 512>>>(in, out, n, clk); });
    timeit("read 4 B", samples * 4, [&] { k_r<<<rows, 512>>>(in, (float*)out, n); });
    timeit("write 8 B", samples * 8, [&] { k_w<<<rows, 512>>>(out, n); });
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("core clock under the read+write kernel: %.2f GHz (%llu shader ticks in %llu x 10 ns)\n", h[0] / (h[1] * 10.0), h[0], h[1]);
    return 0;
}
