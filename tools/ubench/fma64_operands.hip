// Micro-benchmark: v_fma_f64 issue rate and dependent latency on gfx950 by operand kind.
//   vvv : all three sources in VGPRs (lane = channel layout of K1: per-lane coefficients)
//   svv : the multiplier in an SGPR pair (row-owning layout of the fused kernel: wave-uniform coefficients)
//   dep : one accumulator per wave (dependent chain: latency), else 8 independent accumulators (throughput)
// Inline asm so the operand kinds are what the label says.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int ACC>
__global__ void k(double* out, int iters, double a_s, double b_s) {
    double acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = (double)(threadIdx.x + i) * 1e-3;
    // per-lane copies for the vvv form
    double av = a_s + 1e-12 * threadIdx.x, bv = b_s + 1e-15 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / (ACC > 8 ? 8 : 1); ++u) {
#pragma unroll
            for (int i = 0; i < ACC; ++i) {
                if (MODE == 0)
                    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(acc[i]), "v"(av), "v"(bv));
                else if (MODE == 1)
                    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(acc[i]), "s"(a_s), "v"(bv));
                else   // svv with the SGPR as the addend-side multiplier: acc = s * v + acc
                    asm volatile("v_fma_f64 %0, %2, %3, %1" : "=v"(acc[i]) : "v"(acc[i]), "s"(a_s), "v"(bv));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int ACC>
void run(const char* name, int waves_per_simd) {
    const int cus = 256, iters = 4000;
    const int blocks = cus * 4 * waves_per_simd;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, ACC><<<blocks, 64>>>(d, 50, 0.999999, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, ACC><<<blocks, 64>>>(d, iters, 0.999999, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 8 * ACC;
    const double tflops = 2.0 * instr_per_wave * 64 * blocks / (ms * 1e-3) / 1e12;
    const double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    printf("%-14s ACC=%d waves/SIMD=%d: %8.3f ms %6.1f TFLOP/s %6.2f cyc/instr/SIMD (@2.4 GHz nominal)\n", name, ACC,
           waves_per_simd, ms, tflops, cyc);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0, 8>("vvv indep", w);
        run<1, 8>("svv indep", w);
        run<2, 8>("s*v+acc indep", w);
    }
    for (int w : {1, 2, 4}) {
        run<0, 1>("vvv dep", w);
        run<1, 1>("svv dep", w);
        run<0, 2>("vvv 2chains", w);
        run<0, 4>("vvv 4chains", w);
    }
    return 0;
}
