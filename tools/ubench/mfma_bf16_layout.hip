// Operand / result layout of v_mfma_f32_32x32x16_bf16 on gfx950, checked against a host product:
//   A (32 x 16): lane l holds A[l % 32][8 (l / 32) + e], e = 0..7      B (16 x 32): lane l holds B[8 (l / 32) + e][l % 32]
//   D (32 x 32): lane l, register q holds D[(q & 3) + 8 (q >> 2) + 4 (l / 32)][l % 32]   (as v_mfma_f32_32x32x2_f32)
// and the split-bf16 product a1 b1 + a1 b2 + a2 b1 against the float64 product of the float32 operands.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* D1, float* D3) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    bf16x8 a1, a2, b1, b2;
    for (int e = 0; e < 8; ++e) {
        const float a = A[i * 16 + 8 * h + e], b = B[(8 * h + e) * 32 + i];
        a1[e] = (__bf16)a;
        a2[e] = (__bf16)(a - (float)a1[e]);
        b1[e] = (__bf16)b;
        b2[e] = (__bf16)(b - (float)b1[e]);
    }
    f32x16 z = {0};
    f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, z, 0, 0, 0);
    f32x16 d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, z, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, d3, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d3, 0, 0, 0);
    for (int q = 0; q < 16; ++q) {
        const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
        D1[m * 32 + i] = d1[q];
        D3[m * 32 + i] = d3[q];
    }
}

static float bf16r(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u;
    float r;
    memcpy(&r, &u, 4);
    return r;
}

int main() {
    float hA[32 * 16], hB[16 * 32], hD1[1024], hD3[1024];
    srand(5);
    for (float& v : hA) v = (float)rand() / RAND_MAX * 2 - 1;
    for (float& v : hB) v = (float)rand() / RAND_MAX * 2 - 1;
    float *dA, *dB, *dD1, *dD3;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD1, sizeof hD1); hipMalloc(&dD3, sizeof hD3);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD1, dD3);
    hipMemcpy(hD1, dD1, sizeof hD1, hipMemcpyDeviceToHost);
    hipMemcpy(hD3, dD3, sizeof hD3, hipMemcpyDeviceToHost);
    double e1 = 0, e3 = 0, mx = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double r1 = 0, r = 0;
            for (int kk = 0; kk < 16; ++kk) {
                r1 += (double)bf16r(hA[m * 16 + kk]) * bf16r(hB[kk * 32 + n]);
                r += (double)hA[m * 16 + kk] * hB[kk * 32 + n];
            }
            e1 = fmax(e1, fabs(hD1[m * 32 + n] - r1));
            e3 = fmax(e3, fabs(hD3[m * 32 + n] - r));
            mx = fmax(mx, fabs(r));
        }
    printf("layout check: max |D(a1 b1) - host product of the rounded operands| = %.3e (max |D| %.3f)\n", e1, mx);
    printf("split product a1 b1 + a1 b2 + a2 b1 against the exact product of the float32 operands: %.3e\n", e3);
    return e1 < 1e-5 ? 0 : 1;
}
