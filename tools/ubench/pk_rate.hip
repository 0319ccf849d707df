// Micro-benchmark: v_pk_fma_f32 vs v_fma_f32 issue rate on gfx950 at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int ACC>
__global__ void kpk(v2f* out, int iters, v2f a, v2f b) {
    v2f acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = (v2f){(float)(threadIdx.x + i), (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], a, b);
    }
    v2f s = {0, 0};
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ACC>
__global__ void ksc(float* out, int iters, float a, float b) {
    float acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    const int iters = 40000;
    void* d; (void)hipMalloc(&d, 8 * 256 * 4 * 8 * 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * w;
        for (int pk = 0; pk < 2; ++pk) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (pk) kpk<8><<<blocks, 64>>>((v2f*)d, iters, (v2f){1.0000001f, 0.9999999f}, (v2f){1e-9f, 1e-9f});
                else ksc<16><<<blocks, 64>>>((float*)d, iters, 1.0000001f, 1e-9f);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)iters * (pk ? 8 : 16);
            const double flops = 2.0 * (pk ? 2 : 1) * instr * 64 * blocks;
            printf("%s waves/SIMD=%d: %.3f ms %.1f TFLOP/s  %.2f cycles/instr/SIMD(@2.4GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", w, ms,
                   flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (instr * w));
        }
    }
    return 0;
}
