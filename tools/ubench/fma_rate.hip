// Micro-benchmark: issue rate of v_fma_f64 / v_fma_f32 on gfx950 at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int ACC>
__global__ void k(T* out, int iters, T a, T b) {
    T acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = (T)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename T, int ACC>
void run(const char* name, int waves_per_simd) {
    const int cus = 256, iters = 20000;
    const int blocks = cus * 4 * waves_per_simd;   // 64-thread blocks
    T* d; hipMalloc(&d, sizeof(T) * blocks * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<T, ACC><<<blocks, 64>>>(d, 100, (T)1.0000001, (T)1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T, ACC><<<blocks, 64>>>(d, iters, (T)1.0000001, (T)1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * ACC;
    const double tflops = 2.0 * instr_per_wave * 64 * blocks / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at a nominal 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    printf("%s ACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  %.2f cycles/instr/SIMD(@2.4GHz)\n", name, ACC, waves_per_simd, ms, tflops, cyc);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) { run<double, 8>("f64", w); run<double, 2>("f64", w); run<double, 1>("f64", w); }
    for (int w : {1, 2, 4}) { run<float, 8>("f32", w); run<float, 2>("f32", w); run<float, 1>("f32", w);}
    return 0;
}
