// Micro-benchmark: HBM write rate of the filterbank kernel's store pattern - one wave owns 64 rows (channels) of a (C, N) float64
// matrix and walks them in time, a store instruction writing RUN contiguous bytes of 512 / RUN rows - as a function of RUN
// (128 = the shipped tile: one 128-byte line per row and step), with the rows N x 8 bytes apart. No arithmetic. 256 utterances x
// 128 channels x 16000 samples (cfg2: 4.19 GB), two waves per workgroup as in k_erb_filterbank<.., double>.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RUN>
__global__ __launch_bounds__(128) void k_tiles(double* __restrict__ out, int n, int tiles_per_step) {
    constexpr int LPR = RUN / 8;            // lanes per row
    constexpr int RPS = 64 / LPR;           // rows per store instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t unit = (size_t)blockIdx.x * 2 + wave;          // 64 rows
    double* base = out + unit * 64 * (size_t)n;
    const int r0 = lane / LPR, c0 = lane % LPR;
    const int steps = n * 8 / RUN;                               // runs per row
    for (int s = 0; s < steps; s += tiles_per_step) {
        // one tile = 64 rows x RUN bytes = 64 / RPS store instructions; `tiles_per_step` tiles back to back (as after a longer LDS tile)
        for (int t = 0; t < tiles_per_step && s + t < steps; ++t)
#pragma unroll
            for (int q = 0; q < 64 / RPS; ++q) {
                const int row = q * RPS + r0;
                __builtin_nontemporal_store((double)(s + q), base + (size_t)row * n + (size_t)(s + t) * LPR + c0);
            }
        // (the filterbank kernel computes RUN / 8 samples of 64 channels between tiles: ~80 cycles per sample)
        for (int z = 0; z < tiles_per_step * (RUN / 128); ++z) __builtin_amdgcn_s_sleep(20);   // 64 x 20 cycles per 16 samples
    }
}

int main() {
    const int U = 256, C = 128, n = 16000;
    double* out;
    const size_t bytes = (size_t)U * C * n * 8;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
    };
    const dim3 grid(U * C / 64 / 2), block(128);
    run("run 128 B (shipped), one tile at a time", [&] { hipLaunchKernelGGL(k_tiles<128>, grid, block, 0, 0, out, n, 1); });
    run("run 128 B, two tiles back to back", [&] { hipLaunchKernelGGL(k_tiles<128>, grid, block, 0, 0, out, n, 2); });
    run("run 128 B, four tiles back to back", [&] { hipLaunchKernelGGL(k_tiles<128>, grid, block, 0, 0, out, n, 4); });
    run("run 256 B", [&] { hipLaunchKernelGGL(k_tiles<256>, grid, block, 0, 0, out, n, 1); });
    run("run 512 B", [&] { hipLaunchKernelGGL(k_tiles<512>, grid, block, 0, 0, out, n, 1); });
    return 0;
}
