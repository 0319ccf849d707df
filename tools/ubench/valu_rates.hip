// Issue rate of the VALU instructions the bf16 split is made of, one wave per SIMD and two: 16 independent instructions of one kind
// per loop trip (inline asm, so that the compiler neither folds nor vectorises them), cycles per instruction of a wave.
// Question: is v_cvt_pk_bf16_f32 a full-rate instruction (4 cycles per wave64), and what do the integer forms of the split cost?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP16(S) S S S S S S S S S S S S S S S S
template <int KIND>
__global__ __launch_bounds__(512) void k(int waves, int n, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[16];
    unsigned y[16];
    for (int j = 0; j < 16; ++j) {
        x[j] = lane * 0.01f + j;
        y[j] = lane * 77u + j;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < waves) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y[j]) : "v"(x[j]), "v"(x[(j + 1) & 15]));
                if (KIND == 1) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(x[j]) : "v"(x[j]), "v"(x[(j + 1) & 15]));
                if (KIND == 2) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(y[j]) : "v"(y[j]), "v"(y[(j + 1) & 15]), "v"(0x07060302u));
                if (KIND == 3) asm volatile("v_and_b32_e32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(y[(j + 1) & 15]));
                if (KIND == 4) asm volatile("v_max_i32_e32 %0, 0, %1" : "=v"(y[j]) : "v"(y[(j + 1) & 15]));
                if (KIND == 5) asm volatile("v_add_u32_e32 %0, 0x8000, %1" : "=v"(y[j]) : "v"(y[(j + 1) & 15]));
                if (KIND == 6) asm volatile("v_lshlrev_b32_e32 %0, 16, %1" : "=v"(y[j]) : "v"(y[(j + 1) & 15]));
                if (KIND == 7) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*reinterpret_cast<double*>(&x[2 * (j & 7)])) : "v"(*reinterpret_cast<double*>(&x[2 * (j & 7)])), "v"(*reinterpret_cast<double*>(&x[2 * ((j + 1) & 7)])));
                if (KIND == 8) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(y[j]) : "v"(y[j]), "v"(0xffff0000u), "v"(y[(j + 1) & 15]));
                if (KIND == 9) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[j]) : "v"(x[j]), "v"(x[(j + 1) & 15]), "v"(x[(j + 2) & 15]));
                if (KIND == 10) asm volatile("v_cvt_f32_bf16 %0, %1" : "=v"(x[j]) : "v"(y[(j + 1) & 15]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int j = 0; j < 16; ++j) r += x[j] + (float)y[j];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc, std::vector<unsigned long long>& h) {
    const int G = 256, N = 500;
    printf("%-22s", name);
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(G), dim3(512), 0, 0, waves, N, out, cyc);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * G * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < G; ++b)
            for (int w = 0; w < waves; ++w) v.push_back((double)h[b * 8 + w]);
        std::sort(v.begin(), v.end());
        const double c = v[v.size() / 2] / (16.0 * N);
        printf("  %d wave(s) per SIMD: %5.2f cycles per instruction of a wave (%5.2f per SIMD)", waves / 4, c, c / (waves / 4));
    }
    printf("\n");
}

int main() {
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * 512);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8);
    std::vector<unsigned long long> h(256 * 8);
    run<0>("v_cvt_pk_bf16_f32", out, cyc, h);
    run<10>("v_cvt_f32_bf16", out, cyc, h);
    run<1>("v_add_f32", out, cyc, h);
    run<9>("v_fma_f32", out, cyc, h);
    run<7>("v_pk_add_f32", out, cyc, h);
    run<2>("v_perm_b32", out, cyc, h);
    run<3>("v_and_b32 (literal)", out, cyc, h);
    run<8>("v_and_or_b32", out, cyc, h);
    run<4>("v_max_i32", out, cyc, h);
    run<5>("v_add_u32 (literal)", out, cyc, h);
    run<6>("v_lshlrev_b32", out, cyc, h);
    return 0;
}
