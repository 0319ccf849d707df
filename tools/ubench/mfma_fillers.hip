// One instruction stream per wave: v_mfma_f32_32x32x16_bf16 with F independent VALU fillers (v_fma_f32 / v_cvt_pk_bf16_f32 mix)
// after each, on one wave per SIMD and on two (512-thread workgroup: waves w and w + 4 share a SIMD). Cycles per MFMA of a wave
// and of the SIMD: how many fillers hide behind the matrix pipe when they sit in the SAME wave's stream (a partner wave's VALU
// gets only ~2 issue slots per MFMA: mfma_valu_coissue.hip).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int F>
__global__ __launch_bounds__(512) void k(int waves, int n, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (lane + e));
        b[e] = (__bf16)(0.002f * (lane - e));
    }
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = lane * 0.01f + j;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < waves) {
        for (int it = 0; it < n; it += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = MFMA16(a, b, acc);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    if ((f & 3) == 3) {
                        bf16x2 p = {(__bf16)x[f & 7], (__bf16)x[(f + 1) & 7]};
                        x[(f + 2) & 7] += (float)p[0];
                    } else {
                        x[f & 7] = __builtin_fmaf(x[f & 7], 1.0001f, 0.5f);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc[0] + acc[5];
    for (int j = 0; j < 8; ++j) r += x[j];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int F>
void run(float* out, unsigned long long* cyc, std::vector<unsigned long long>& h) {
    const int G = 256, N = 2000;
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<F>, dim3(G), dim3(512), 0, 0, waves, N, out, cyc);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * G * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int g = 0; g < G; ++g)
            for (int w = 0; w < waves; ++w) v.push_back((double)h[g * 8 + w]);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        printf("F = %2d fillers per MFMA, %d wave(s) per SIMD: %.1f cycles per MFMA of a wave = %.1f per MFMA of the SIMD, %.2f cycles per instruction\n",
               F, waves / 4, med / N, med / N / (waves / 4), med / N / (waves / 4) / (F + 1));
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * 512);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8);
    std::vector<unsigned long long> h(256 * 8);
    run<0>(out, cyc, h);
    run<2>(out, cyc, h);
    run<4>(out, cyc, h);
    run<6>(out, cyc, h);
    run<8>(out, cyc, h);
    run<12>(out, cyc, h);
    run<16>(out, cyc, h);
    return 0;
}
