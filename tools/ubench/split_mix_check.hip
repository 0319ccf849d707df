#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4_c(const float v0, const float v1, const float v2, const float v3, h16x4& hi, h16x4& lo) {
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1, h2 = (_Float16)v2, h3 = (_Float16)v3;
    hi = h16x4{h0, h1, h2, h3};
    lo = h16x4{(_Float16)(v0 - (float)h0), (_Float16)(v1 - (float)h1), (_Float16)(v2 - (float)h2), (_Float16)(v3 - (float)h3)};
}
// hi pair by one packed conversion, lo pieces by mixed-precision FMAs that read the fp16 halves directly and round x - hi to fp16
__device__ __forceinline__ void split2_mix(float v0, float v1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
}
__device__ __forceinline__ void split4_mix(const float v0, const float v1, const float v2, const float v3, h16x4& hi, h16x4& lo) {
    u32x2 h, l;
    unsigned a, b;
    split2_mix(v0, v1, a, b); h[0] = a; l[0] = b;
    split2_mix(v2, v3, a, b); h[1] = a; l[1] = b;
    hi = __builtin_bit_cast(h16x4, h);
    lo = __builtin_bit_cast(h16x4, l);
}
__device__ __forceinline__ void split2_scaled(float v0, float v1, float c, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(v0), "v"(c));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(v1), "v"(c));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(v0), "v"(c), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(v1), "v"(c), "v"(hi));
}
__global__ void k_scaled(const float4* x, h16x4* o, int n, float c) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = x[i];
    h16x4 hi, lo;
    split4_c(v.x * c, v.y * c, v.z * c, v.w * c, hi, lo);
    u32x2 h, l;
    unsigned a, b;
    split2_scaled(v.x, v.y, c, a, b); h[0] = a; l[0] = b;
    split2_scaled(v.z, v.w, c, a, b); h[1] = a; l[1] = b;
    o[4 * i] = hi; o[4 * i + 1] = lo; o[4 * i + 2] = __builtin_bit_cast(h16x4, h); o[4 * i + 3] = __builtin_bit_cast(h16x4, l);
}
__global__ void k(const float4* x, h16x4* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = x[i];
    h16x4 hi, lo, hi2, lo2;
    split4_c(v.x, v.y, v.z, v.w, hi, lo);
    split4_mix(v.x, v.y, v.z, v.w, hi2, lo2);
    o[4 * i] = hi; o[4 * i + 1] = lo; o[4 * i + 2] = hi2; o[4 * i + 3] = lo2;
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(sizeof(float) * 4 * n);
    srand(1);
    for (int i = 0; i < 4 * n; ++i) {
        unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand();
        int mode = i % 5;
        float v;
        if (mode == 0) { memcpy(&v, &r, 4); if (!(fabsf(v) < 60000.f)) v = 0.f; }         // any bit pattern below the fp16 maximum
        else if (mode == 1) v = (float)(r % 100000) / 100000.f * 16384.f;                  // the scaled activations' range
        else if (mode == 2) v = ldexpf((float)(r & 0xffffff) / 16777216.f, -(int)(r >> 27) - 8);   // small, into fp16 subnormals
        else if (mode == 3) v = -(float)(r % 4096) * 0.37f;
        else v = (float)(r & 0xfff) * 0.5f + 0.000244140625f * (float)(r >> 28);           // near fp16 ties
        h[i] = v;
    }
    float4* dx; h16x4* dout;
    hipMalloc(&dx, sizeof(float) * 4 * n); hipMalloc(&dout, sizeof(h16x4) * 4 * n);
    hipMemcpy(dx, h, sizeof(float) * 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    unsigned short* out = (unsigned short*)malloc(sizeof(h16x4) * 4 * n);
    hipMemcpy(out, dout, sizeof(h16x4) * 4 * n, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 8; ++j)
            if (out[16 * i + j] != out[16 * i + 8 + j]) { if (bad < 5) printf("mismatch at %d/%d: %04x vs %04x (x = %g)\n", i, j, out[16 * i + j], out[16 * i + 8 + j], h[4 * i + (j & 3)]); ++bad; }
    printf("%ld of %d pieces differ between the C split and the v_fma_mix split\n", bad, 8 * n);
    long bad2 = 0;
    for (float c : {0.0009765625f, 0.125f, 1.f, 4.f}) {
        hipLaunchKernelGGL(k_scaled, dim3(n / 256), dim3(256), 0, 0, dx, dout, n, c);
        hipMemcpy(out, dout, sizeof(h16x4) * 4 * n, hipMemcpyDeviceToHost);
        long b2 = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < 8; ++j)
                if (out[16 * i + j] != out[16 * i + 8 + j]) { if (b2 < 3) printf("scaled (c = %g) mismatch at %d/%d: %04x vs %04x (x = %g)\n", c, i, j, out[16 * i + j], out[16 * i + 8 + j], h[4 * i + (j & 3)]); ++b2; }
        printf("c = %g: %ld of %d pieces differ between split(v * c) in C and the form with the multiplication inside the v_fma_mix instructions (not used: same speed; differs in the sign of zeros)\n", c, b2, 8 * n);
        bad2 += b2;
    }
    return bad != 0;
}
