// Probe: cycles per MFMA (s_memtime) for the instructions the policy kernel issues, one wavefront per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>   // MODE 0: f32 32x32x2, 1: i8 32x32x32, 2: f32 with a VALU op feeding B between pairs
__global__ __launch_bounds__(256) void probe(unsigned long long *out, float seed) {
    v16f acc[4]; v16i iacc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; iacc[i][r] = 0; }
    float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
    v4i ia = {(int)threadIdx.x, 1, 2, 3}, ib = {4, 5, 6, (int)threadIdx.x};
    const int iters = 512;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 1) iacc[u % NACC] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, iacc[u % NACC], 0, 0, 0);
            else {
                if (MODE == 2 && (u & 1) == 0) b = fmaxf(__builtin_fmaf(b, 1.0001f, a), 0.f);
                acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = clock64();
    float s = 0; int si = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) { s += acc[i][r]; si += iacc[i][r]; }
    if (s == 12345.f || si == 12345) out[1000000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int MODE>
void run(const char *name, unsigned long long *d, int blocks) {
    hipMemset(d, 0, 8 * 4096);
    hipLaunchKernelGGL((probe<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < blocks * 4 && i < 1024; ++i) sum += h[i];
    double cyc = sum / (blocks * 4 < 1024 ? blocks * 4 : 1024) / (512.0 * 8);
    printf("%-44s blocks %4d: %.1f cycles/MFMA (s_memtime), kernel %.3f ms -> %.1f ns/MFMA => clock ~%.0f MHz\n", name, blocks, cyc, ms,
           ms * 1e6 / (512.0 * 8), cyc / (ms * 1e6 / (512.0 * 8)) * 1e3);
}

int main() {
    unsigned long long *d; hipMalloc(&d, 8 * 1000008);
    for (int blocks : {1, 256}) {
        run<1, 0>("f32 32x32x2, 1 accumulator", d, blocks);
        run<2, 0>("f32 32x32x2, 2 accumulators", d, blocks);
        run<4, 0>("f32 32x32x2, 4 accumulators", d, blocks);
        run<2, 2>("f32 32x32x2, 2 acc + VALU feeding B per pair", d, blocks);
        run<1, 1>("i8 32x32x32, 1 accumulator", d, blocks);
        run<4, 1>("i8 32x32x32, 4 accumulators", d, blocks);
    }
    return 0;
}
