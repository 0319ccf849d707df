// Micro-benchmark behind DESIGN.md section 7, item 1: would k_envelope<double> (float64 Hilbert FFT, 8192 complex points = 128 KB:
// one workgroup per CU) gain from a second resident workgroup if the real and imaginary parts went through a 64 KB LDS array one
// after the other?  One iteration = one transform pass of a 512-thread workgroup holding 16 complex float64 points per thread:
//   A  one workgroup per CU:  16 ds_write_b128, barrier, 16 ds_read_b128, NV v_fma_f64, barrier              (as shipped)
//   B  two workgroups per CU: 16 ds_write_b64 (re), barrier, 16 ds_read_b64, barrier, 16 ds_write_b64 (im), barrier,
//                             16 ds_read_b64, NV v_fma_f64, barrier                                            (the proposal)
// Reported: microseconds per pass and CU slot (B: iteration time / 2, two rows advance per iteration), for LDS only, VALU only, both.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int NV>
__device__ __forceinline__ void valu_block(double (&acc)[8], double a, double b) {
#pragma unroll
    for (int r = 0; r < NV / 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
}

template <bool SPLIT, int NW, int NV>
__global__ __launch_bounds__(512) void k(double* out, int iters, double a, double b) {
    constexpr int ST = 544;
    __shared__ double lds[(SPLIT ? 1 : 2) * (16 * ST + 64)];
    const int tid = threadIdx.x;
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (double)(tid + i) * 1e-3;
    d2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = {acc[i & 7], -acc[i & 7]};
    const int rd = (tid + 64) & 511;
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < NW; ++i) lds[i * ST + tid] = v[i].x;
            __syncthreads();
            double t[16];
#pragma unroll
            for (int i = 0; i < NW; ++i) t[i] = lds[i * ST + rd];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NW; ++i) lds[i * ST + tid] = v[i].y;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                v[i].y = lds[i * ST + rd] + 1e-300 * t[i];
                v[i].x = t[i];
            }
        } else {
            d2* l2 = reinterpret_cast<d2*>(lds);
#pragma unroll
            for (int i = 0; i < NW; ++i) l2[i * ST + tid] = v[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NW; ++i) v[i] = l2[i * ST + rd];
        }
        valu_block<NV>(acc, a, b);
        __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 512 + tid] = s;
}

template <bool SPLIT, int NW, int NV>
float run() {
    const int iters = 2000;
    const int blocks = 256 * (SPLIT ? 2 : 1);
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<SPLIT, NW, NV><<<blocks, 512>>>(d, 50, 0.999, 1e-6);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SPLIT, NW, NV><<<blocks, 512>>>(d, iters, 0.999, 1e-6);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms * 1e3f / iters / (SPLIT ? 2 : 1);
}

int main() {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<true, 16, 320>, 512, 0);
    printf("resident workgroups per CU: split %d", nb);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<false, 16, 320>, 512, 0);
    printf(", whole %d\n", nb);
    printf("A one workgroup per CU, 16-byte exchanges:   LDS only %6.3f us  VALU only (NV=320) %6.3f us  both %6.3f us | NV=640: VALU %6.3f both %6.3f\n",
           run<false, 16, 0>(), run<false, 0, 320>(), run<false, 16, 320>(), run<false, 0, 640>(), run<false, 16, 640>());
    printf("B two workgroups per CU, re / im separately: LDS only %6.3f us  VALU only (NV=320) %6.3f us  both %6.3f us | NV=640: VALU %6.3f both %6.3f\n",
           run<true, 16, 0>(), run<true, 0, 320>(), run<true, 16, 320>(), run<true, 0, 640>(), run<true, 16, 640>());
    return 0;
}
