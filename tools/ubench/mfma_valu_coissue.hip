// Two waves on one SIMD (512-thread workgroup, one per CU): wave A issues a stream of v_mfma_f32_32x32x16_bf16, its partner B
// (wave A + 4: same SIMD) a stream of VALU / LDS / global-store work. How long does B's work take beside A's stream, and A's
// beside B - with A's MFMAs on ONE accumulator (a dependent chain) or alternating between two or three?
//   mode bits: 1 = waves 0-3 run the MFMA stream, 2 = waves 4-7 run the VALU stream; chains = 1, 2, 3; kind = 0 v_fma chain
//   (dependent), 1 independent v_fma x 4, 2 ds_read_b128 + v_add, 3 v_cvt_pk / v_sub mix, 4 MFMA (three per 40 VALU)
// Output: cycles (s_memtime) per wave for each, median over workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

template <int CHAINS>
__device__ __forceinline__ void mfma_stream(int n, f32x16* out, int lane) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (lane + e));
        b[e] = (__bf16)(0.002f * (lane - e));
    }
    f32x16 acc[3];
    for (int c = 0; c < 3; ++c)
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    for (int it = 0; it < n; it += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[u % CHAINS] = MFMA16(a, b, acc[u % CHAINS]);
    }
    f32x16 r = acc[0];
    if (CHAINS > 1) r += acc[1];
    if (CHAINS > 2) r += acc[2];
    out[lane] = r;
}

__device__ __forceinline__ void valu_stream(int kind, int n, float* out, int lane, float* lds) {
    float x0 = lane * 0.5f, x1 = lane * 0.25f, x2 = 1.f + lane, x3 = 2.f - lane;
    if (kind == 0) {
        for (int it = 0; it < n; it += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) x0 = __builtin_fmaf(x0, 1.0001f, 0.5f);
        }
    } else if (kind == 1) {
        for (int it = 0; it < n; it += 8) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                x0 = __builtin_fmaf(x0, 1.0001f, 0.5f);
                x1 = __builtin_fmaf(x1, 1.0002f, 0.25f);
                x2 = __builtin_fmaf(x2, 0.9999f, 0.125f);
                x3 = __builtin_fmaf(x3, 0.9998f, 0.75f);
            }
        }
    } else if (kind == 2) {
        const float4* p = reinterpret_cast<const float4*>(lds) + lane;
        for (int it = 0; it < n; it += 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 v = p[64 * ((it + u) & 7)];
                x0 += v.x;
                asm volatile("" : "+v"(x0));
            }
        }
    } else if (kind == 3) {
        for (int it = 0; it < n; it += 8) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 hpk = {(__bf16)x0, (__bf16)x1};
                const float h0 = (float)hpk[0], h1 = (float)hpk[1];
                x2 += x0 - h0;
                x3 += x1 - h1;
                asm volatile("" : "+v"(x2), "+v"(x3));
                x0 += 1.f;
                x1 += 2.f;
            }
        }
    } else {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)(0.001f * (lane + e));
            b[e] = (__bf16)(0.002f * (lane - e));
        }
        f32x16 acc;
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        for (int it = 0; it < n; it += 40) {
            acc = MFMA16(a, b, acc);
            acc = MFMA16(a, b, acc);
            acc = MFMA16(a, b, acc);
#pragma unroll
            for (int u = 0; u < 37; ++u) x0 = __builtin_fmaf(x0, 1.0001f, acc[u & 15] * 1e-30f);
        }
        x1 += acc[3];
    }
    out[lane] = x0 + x1 + x2 + x3;
}

template <int CHAINS>
__global__ __launch_bounds__(512) void k(int mode, int kind, int nmfma, int nvalu, f32x16* mo, float* vo, unsigned long long* cyc) {
    __shared__ float lds[64 * 4 * 8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < 64 * 4 * 8; e += 512) lds[e] = e * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) mfma_stream<CHAINS>(nmfma, mo + (blockIdx.x * 8 + wave) * 64, lane);
    } else {
        if (mode & 4) __builtin_amdgcn_s_setprio(3);       // the VALU wave outranks the MFMA wave
        if (mode & 2) valu_stream(kind, nvalu, vo + (blockIdx.x * 8 + wave) * 64, lane, lds);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int G = 256;
    f32x16* mo;
    float* vo;
    unsigned long long* cyc;
    hipMalloc(&mo, sizeof(f32x16) * G * 8 * 64);
    hipMalloc(&vo, sizeof(float) * G * 8 * 64);
    hipMalloc(&cyc, sizeof(unsigned long long) * G * 8);
    std::vector<unsigned long long> h(G * 8);
    auto run = [&](int chains, int mode, int kind, int nm, int nv, double& ma, double& va) {
        for (int rep = 0; rep < 3; ++rep) {
            if (chains == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(512), 0, 0, mode, kind, nm, nv, mo, vo, cyc);
            else if (chains == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(512), 0, 0, mode, kind, nm, nv, mo, vo, cyc);
            else hipLaunchKernelGGL(k<3>, dim3(G), dim3(512), 0, 0, mode, kind, nm, nv, mo, vo, cyc);
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * G * 8, hipMemcpyDeviceToHost);
        std::vector<double> a, b;
        for (int g = 0; g < G; ++g)
            for (int w = 0; w < 8; ++w) (w < 4 ? a : b).push_back((double)h[g * 8 + w]);
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        ma = a[a.size() / 2];
        va = b[b.size() / 2];
    };
    const int NM = 60 * 20, NV = 4000;
    const char* kinds[] = {"dependent v_fma chain", "4 independent v_fma chains", "ds_read_b128 + v_add", "cvt_pk/sub split mix", "3 MFMA per 40 VALU"};
    for (int chains = 1; chains <= 2; ++chains) {
        double ma, va, m2, v2;
        run(chains, 1, 0, NM, NV, ma, va);
        printf("MFMA stream alone, %d chain(s): %.0f cycles for %d MFMAs = %.1f per MFMA\n", chains, ma, NM, ma / NM);
        for (int kind = 0; kind < 5; ++kind) {
            run(chains, 2, kind, NM, NV, m2, va);
            run(chains, 3, kind, NM, NV, m2, v2);
            printf("  partner: %-28s alone %.0f cycles (%.2f per op), beside the MFMA stream %.0f (%.2f per op); MFMA stream then %.0f (%.1f per MFMA)\n",
                   kinds[kind], va, va / NV, v2, v2 / NV, m2, m2 / NM);
            if (chains == 1) {
                run(chains, 7, kind, NM, NV, m2, v2);
                printf("           with s_setprio 3 on the partner: %.0f (%.2f per op); MFMA stream then %.0f (%.1f per MFMA)\n", v2, v2 / NV, m2, m2 / NM);
            }
        }
    }
    return 0;
}
