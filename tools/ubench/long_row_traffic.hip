// Micro-benchmark: the MEMORY SIDE of a one-kernel (spectral) route for rows of 65 281 .. 131 072 samples (131 072-point
// transforms = eight 16384-point sub-transforms), with no arithmetic at all - what the fabric gives such a row when every
// CU works on one ("a measured prototype, not a paper estimate"). A row is a (utterance, channel) pair;
// 128 channels; the channel tables are 65 544 x {H, u} = 1 MB each (134 MB for the bank), an utterance's spectrum 512 KB.
//
//   A  "decimated": sub-transform j takes the bins 8 k + j (its slice of the table and of the spectrum, each read once), writes
//      its 16384 complex outputs (128 KB) to a parking area; a last sweep reads the eight parked planes (1 MB), combines and
//      stores the float64 envelopes. Per row: 1.5 MB read + 1 MB parked + 1 MB read back + 8 n bytes stored.
//   B  "stash": the bins are formed once (1.5 MB read) and kept as a 512 KB stash; each of the eight sub-transforms (which here
//      produce every eighth sample, so their magnitudes can be taken at once) reads the whole stash; magnitudes parked as float
//      (4 n bytes), read back by the low-pass sweep, 8 n bytes stored. Per row: 1.5 + 4 MB read, 0.5 MB stash, 8 n + 8 n bytes.
//
// One persistent 512-thread workgroup per CU, rows in channel-major order (a launch's concurrent rows share ~9 tables and all
// spectra), 16-byte loads, 16-byte non-temporal envelope stores. Prints microseconds per row (256 rows in flight).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BINS = 65536;         // (one more in the real thing)
constexpr int SUB = 16384;

__device__ __forceinline__ f4 ld16(const f4* p) { return *p; }

template <int VARIANT>
__global__ __launch_bounds__(512) void k_rows(const f4* __restrict__ tables, const f4* __restrict__ spectra, f4* __restrict__ park,
                                              double* __restrict__ env, int U, int C, int n, int rows) {
    const int tid = threadIdx.x;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
        const int c = r / U, u = r - c * U;
        const f4* tab = tables + (size_t)c * BINS;                 // 16 bytes per bin
        const f4* spc = spectra + (size_t)u * (BINS / 2);         // 8 bytes per bin
        f4* pk = park + (size_t)(blockIdx.x) * (VARIANT == 0 ? 8 * SUB / 2 : BINS / 2 + SUB * 8 / 4);   // a parking area per workgroup
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        if (VARIANT == 0) {
            for (int j = 0; j < 8; ++j) {
                // slice j of the table (8192 bins x 16 B = 128 KB) and of the spectrum (64 KB)
                for (int k = tid; k < BINS / 8; k += 512) acc += ld16(tab + (size_t)j * (BINS / 8) + k);
                for (int k = tid; k < BINS / 16; k += 512) acc += ld16(spc + (size_t)j * (BINS / 16) + k);
                // 16384 complex outputs of the sub-transform
                for (int k = tid; k < SUB / 2; k += 512) pk[(size_t)j * (SUB / 2) + k] = acc;
            }
            __syncthreads();
            // combine: sample group m reads one complex value of each plane (here: two per 16-byte load), stores 8 x 2 doubles
            d2* y = reinterpret_cast<d2*>(env + (size_t)r * n);
            for (int k = tid; k < SUB / 2; k += 512) {
                f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) s += ld16(pk + (size_t)j * (SUB / 2) + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int smp = j * SUB + 2 * k;               // samples n1 + 16384 n2
                    if (smp < n) __builtin_nontemporal_store(d2{(double)s[0] + j, (double)s[1]}, y + smp / 2);
                }
            }
        } else {
            f4* stash = pk;                                         // 512 KB
            float* mag = reinterpret_cast<float*>(pk + BINS / 2);   // 4 bytes per sample
            for (int k = tid; k < BINS; k += 512) {
                acc += ld16(tab + k);
                if ((k & 1) == 0) acc += ld16(spc + (k >> 1));
                if ((k & 1) == 0) stash[k >> 1] = acc;
            }
            __syncthreads();
            for (int j = 0; j < 8; ++j) {
                f4 s = {0.f, 0.f, 0.f, 0.f};
                for (int k = tid; k < BINS / 2; k += 512) s += ld16(stash + k);
                // 16384 magnitudes of samples 8 m + j (written 4 per lane into plane j)
                for (int k = tid; k < SUB / 4; k += 512) reinterpret_cast<f4*>(mag)[(size_t)j * (SUB / 4) + k] = s;
            }
            __syncthreads();
            d2* y = reinterpret_cast<d2*>(env + (size_t)r * n);
            for (int k = tid; k < SUB / 4; k += 512) {
                f4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) s += ld16(reinterpret_cast<const f4*>(mag) + (size_t)j * (SUB / 4) + k);
                // samples 32 k .. 32 k + 31
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int smp = 32 * k + 2 * q;
                    if (smp < n) __builtin_nontemporal_store(d2{(double)s[q & 3], (double)q}, y + smp / 2);
                }
            }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int U = argc > 1 ? atoi(argv[1]) : 30, C = 128;
    int ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    f4 *tables, *spectra, *park;
    double* env;
    const size_t tab_b = (size_t)C * BINS * 16, spc_b = (size_t)U * BINS * 8, park_b = (size_t)ncu * (1 << 20) + (1 << 20);
    const size_t env_b = (size_t)U * C * 131072 * 8;
    if (hipMalloc(&tables, tab_b) != hipSuccess || hipMalloc(&spectra, spc_b) != hipSuccess || hipMalloc(&park, park_b) != hipSuccess ||
        hipMalloc(&env, env_b) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    hipMemset(tables, 0, tab_b);
    hipMemset(spectra, 0, spc_b);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int rows = U * C;
    printf("%d utterances x %d channels = %d rows, %d workgroups; tables %.0f MB, spectra %.1f MB\n", U, C, rows, ncu, tab_b / 1e6, spc_b / 1e6);
    for (int n : {80000, 131072}) {
        for (int variant = 0; variant < 2; ++variant) {
            auto launch = [&]() {
                if (variant == 0) hipLaunchKernelGGL(k_rows<0>, dim3(ncu), dim3(512), 0, 0, tables, spectra, park, env, U, C, n, rows);
                else hipLaunchKernelGGL(k_rows<1>, dim3(ncu), dim3(512), 0, 0, tables, spectra, park, env, U, C, n, rows);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            const double per_row_us = ms * 1e3 / ((double)rows / ncu);
            const double mb = variant == 0 ? 1.5 + 1.0 + 1.0 + 8.0 * n / 1e6 : 1.5 + 0.5 + 4.0 + 8.0 * n / 1e6 + 8.0 * n / 1e6 * 0.5 * 2;
            printf("n = %6d  variant %s: %.2f ms per launch, %.0f us per row and CU (%.2f MB per row through the fabric: %.2f TB/s chip-wide); "
                   "audio-seconds/s if this were all: %.0f\n",
                   n, variant == 0 ? "A (decimated, complex parking)" : "B (stash + float parking)     ", ms, per_row_us, mb,
                   mb * 1e6 * rows / (ms * 1e-3) / 1e12, (double)U * n / 16000.0 / (ms * 1e-3));
        }
    }
    return 0;
}
