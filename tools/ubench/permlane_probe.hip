// What v_permlane32_swap / v_permlane16_swap do to two registers (lane values 100 + lane and 200 + lane).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(100u + l, 200u + l, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(100u + l, 200u + l, false, false);
    out[128 + l] = s[0];
    out[192 + l] = s[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap r[0]", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]"};
    for (int q = 0; q < 4; ++q) {
        printf("%s:", names[q]);
        for (int l = 0; l < 64; l += 4) printf(" %u", h[64 * q + l]);
        printf("\n");
    }
    return 0;
}
