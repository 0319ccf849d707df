// Micro-benchmark: do LDS exchanges (ds_write_b64 + ds_read_b64, the traffic of one FFT pass: 16 + 16 per thread, 64 KB +
// 64 KB per 512-thread workgroup) and f32 VALU work (NV v_fma_f32 per thread) overlap on a gfx950 CU, and what do the
// workgroup barriers of a pass-structured kernel cost?  One iteration = one "pass".
//   MODE 0  every wave: writes, reads, VALU, no barrier
//   MODE 1  every wave: writes, barrier, reads, VALU, barrier     (the shape of k_envelope's passes)
//   MODE 2  role split, no barrier: waves 0-3 do all the LDS traffic (2x each), waves 4-7 all the VALU (2x each)
//   MODE 3  as MODE 1 but the reads of the NEXT pass are issued before the VALU block of this one (software pipelining
//           across two buffers: two rows in flight per workgroup)
// Reported: microseconds per iteration per workgroup slot, for LDS only / VALU only / both, at 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int NV>
__device__ __forceinline__ void valu_block(float (&acc)[16], float a, float b) {
#pragma unroll
    for (int r = 0; r < NV / 16; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
}

template <int MODE, int NW, int NV, int LDS_KB>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    __shared__ f2 lds[LDS_KB * 128 + 1024];   // 64 + 8 KiB -> two workgroups per CU, 128 + 8 -> one
    constexpr int ST = 544;   // row pitch of k_envelope's padded array (keeps the compiler from pairing the accesses)
    const int tid = threadIdx.x;
    const int wv = tid >> 6;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(tid + i) * 1e-3f;
    f2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = {acc[i], -acc[i]};
    const int rd = (tid + 64) & 511;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            if (wv < 4) {
                // all of the workgroup's LDS traffic on half the waves
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < NW; ++i) lds[i * ST + ((tid + h * 256) & 511)] = v[i];
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        const f2 t = lds[i * ST + ((rd + h * 256) & 511)];
                        v[i].x += t.y * 1e-30f;
                    }
                }
            } else {
                valu_block<NV>(acc, a, b);
                valu_block<NV>(acc, a, b);
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) lds[i * ST + tid] = v[i];
        if (MODE == 1 || MODE == 3) __syncthreads();
        f2 t[16];
#pragma unroll
        for (int i = 0; i < NW; ++i) t[i] = lds[i * ST + rd];
        if (MODE == 3) {
            // VALU block first (independent of t), the loaded values are consumed after it
            valu_block<NV>(acc, a, b);
#pragma unroll
            for (int i = 0; i < NW; ++i) v[i].x += t[i].y * 1e-30f;
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) v[i].x += t[i].y * 1e-30f;
            valu_block<NV>(acc, a, b);
        }
        if (MODE == 1 || MODE == 3) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i].x;
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int NW, int NV, int LDS_KB>
float run(int wgs_per_cu) {
    const int iters = 4000;
    const int blocks = 256 * wgs_per_cu;
    float* d;
    hipMalloc(&d, sizeof(float) * blocks * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, NW, NV, LDS_KB><<<blocks, 512>>>(d, 100, 0.999f, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NW, NV, LDS_KB><<<blocks, 512>>>(d, iters, 0.999f, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms * 1e3f / iters;   // us per iteration (all workgroups of a CU run concurrently)
}

template <int MODE>
void table(const char* name) {
    printf("%s\n", name);
    printf("  1 WG/CU (128 KB):  LDS only %6.3f us  VALU only %6.3f us  both %6.3f us   (NV=320)\n", run<MODE, 16, 0, 128>(1),
           run<MODE, 0, 320, 128>(1), run<MODE, 16, 320, 128>(1));
    printf("  2 WG/CU ( 64 KB):  LDS only %6.3f us  VALU only %6.3f us  both %6.3f us   (NV=320; per pair of passes)\n",
           run<MODE, 16, 0, 64>(2), run<MODE, 0, 320, 64>(2), run<MODE, 16, 320, 64>(2));
    printf("  2 WG/CU ( 64 KB):  LDS only %6.3f us  VALU only %6.3f us  both %6.3f us   (NV=160)\n", run<MODE, 16, 0, 64>(2),
           run<MODE, 0, 160, 64>(2), run<MODE, 16, 160, 64>(2));
}

int main() {
    table<0>("MODE 0: every wave writes, reads, computes; no barrier");
    table<1>("MODE 1: write, barrier, read, compute, barrier");
    table<3>("MODE 3: write, barrier, read issued, compute, consume, barrier");
    table<2>("MODE 2: role split (waves 0-3 LDS x2, waves 4-7 VALU x2), no barrier");
    return 0;
}
