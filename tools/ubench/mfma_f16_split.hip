// v_mfma_f32_32x32x16_f16 on gfx950 as the engine of a split-FP16 product: operand layout (expected:
// that of the bf16 instruction), the split product a1 b1 + a1 b2 + a2 b1 with a = a1 + a2 in two fp16 pieces against the float64
// product of the float32 operands - for operands of order 1, for operands scaled into fp16's comfortable range, and for
// operands whose low pieces are fp16 subnormals (does the matrix core keep them?) - and the issue rate beside the bf16 form.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float sa, float sb, float* D1, float* D3) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    h16x8 a1, a2, b1, b2;
    for (int e = 0; e < 8; ++e) {
        const float a = A[i * 16 + 8 * h + e] * sa, b = B[(8 * h + e) * 32 + i] * sb;
        a1[e] = (_Float16)a;
        a2[e] = (_Float16)(a - (float)a1[e]);
        b1[e] = (_Float16)b;
        b2[e] = (_Float16)(b - (float)b1[e]);
    }
    f32x16 z = {0};
    f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, z, 0, 0, 0);
    f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, z, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, d3, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, d3, 0, 0, 0);
    const float inv = 1.0f / (sa * sb);
    for (int q = 0; q < 16; ++q) {
        const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
        D1[m * 32 + i] = d1[q];
        D3[m * 32 + i] = d3[q] * inv;
    }
}

template <bool F16>
__global__ void rate(float* out, int iters) {
    h16x8 ah, bh;
    bf16x8 ab, bb;
    for (int e = 0; e < 8; ++e) {
        ah[e] = (_Float16)(0.001f * (threadIdx.x + e));
        bh[e] = (_Float16)(0.002f * (threadIdx.x + e));
        ab[e] = (__bf16)(0.001f * (threadIdx.x + e));
        bb[e] = (__bf16)(0.002f * (threadIdx.x + e));
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        if (F16) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

static float f16r(float x) { return (float)(_Float16)x; }

static void check(const char* what, float ascale, float bscale, float sa, float sb) {
    float hA[32 * 16], hB[16 * 32], hD1[1024], hD3[1024];
    srand(5);
    for (float& v : hA) v = ((float)rand() / RAND_MAX * 2 - 1) * ascale;
    for (float& v : hB) v = ((float)rand() / RAND_MAX * 2 - 1) * bscale;
    float *dA, *dB, *dD1, *dD3;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD1, sizeof hD1); hipMalloc(&dD3, sizeof hD3);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, sa, sb, dD1, dD3);
    hipMemcpy(hD1, dD1, sizeof hD1, hipMemcpyDeviceToHost);
    hipMemcpy(hD3, dD3, sizeof hD3, hipMemcpyDeviceToHost);
    double e1 = 0, e3 = 0, mx = 0, m1 = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double r1 = 0, r = 0;
            for (int kk = 0; kk < 16; ++kk) {
                r1 += (double)f16r(hA[m * 16 + kk] * sa) * f16r(hB[kk * 32 + n] * sb);
                r += (double)hA[m * 16 + kk] * hB[kk * 32 + n];
            }
            e1 = fmax(e1, fabs(hD1[m * 32 + n] - r1));
            m1 = fmax(m1, fabs(r1));
            e3 = fmax(e3, fabs(hD3[m * 32 + n] - r));
            mx = fmax(mx, fabs(r));
        }
    printf("%-58s layout / single product error %.2e of %.3g; split product error %.3e of max |D| %.3g = %.2e relative\n", what, e1, m1, e3, mx,
           e3 / mx);
    hipFree(dA); hipFree(dB); hipFree(dD1); hipFree(dD3);
}

int main() {
    check("operands of order 1, no scaling:", 1.f, 1.f, 1.f, 1.f);
    check("activations ~0.3, weights ~0.1, no scaling:", 0.3f, 0.1f, 1.f, 1.f);
    check("activations ~0.3, weights ~0.1, scaled 2^10 / 2^14:", 0.3f, 0.1f, 1024.f, 16384.f);
    check("activations ~1e-3 (low pieces subnormal), no scaling:", 1e-3f, 0.1f, 1.f, 1.f);
    check("activations ~1e-3, scaled 2^10 / 2^14:", 1e-3f, 0.1f, 1024.f, 16384.f);
    float* out;
    hipMalloc(&out, 1024 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int f16 = 0; f16 < 2; ++f16) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f16) rate<true><<<1024, 256>>>(out, iters);
            else rate<false><<<1024, 256>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 32 * 32 * 16 * 4.0 * iters * 1024 * 4;
        printf("%s: %.1f TFLOP/s (4 independent accumulators, 4 waves per workgroup, 1024 workgroups)\n",
               f16 ? "v_mfma_f32_32x32x16_f16 " : "v_mfma_f32_32x32x16_bf16", flop / ms / 1e9);
    }
    return 0;
}
