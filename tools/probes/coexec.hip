// Probe: do the two wavefronts of one SIMD co-execute matrix and vector work?  512-thread workgroups (wavefronts w and
// w + 4 share a SIMD), one per CU.  Role A (waves 0-3) issues MFMAs, role B (waves 4-7) issues independent v_fma chains;
// each role is timed alone and with the other one running beside it.  If the kinds of work overlap, "both" costs about
// max(A, B); if they share an execution resource it costs about A + B.
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip ; diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_i32_32x32x32_i8, 2: v_mfma_f32_32x32x16_bf16
template <int KIND>
__global__ __launch_bounds__(512) void probe(float *sink, float seed, int run_a, int run_b, int iters) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!run_a) return;
        v16f acc[2]; v16i iacc[2];
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; iacc[i][r] = 0; }
        float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
        v4i ia = {(int)threadIdx.x, 1, 2, 3}, ib = {4, 5, 6, (int)threadIdx.x};
        v8bf ba, bb;
        for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (KIND == 0) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
                else if (KIND == 1) iacc[u & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, iacc[u & 1], 0, 0, 0);
                else acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[u & 1], 0, 0, 0);
            }
        }
        float s = 0; int si = 0;
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) { s += acc[i][r]; si += iacc[i][r]; }
        if (s == 12345.f || si == 12345) sink[0] = s;
    } else {
        if (!run_b) return;
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = seed + i + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);      // 256 independent-chain v_fma per iteration
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        if (s == 12345.f) sink[1] = s;
    }
}

template <int KIND>
float run(float *d, int a, int b, int iters) {
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(512), 0, 0, d, 1.0f, a, b, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(512), 0, 0, d, 1.0f, a, b, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

template <int KIND>
void report(const char *name, float *d) {
    const int iters = 2000;
    const float ta = run<KIND>(d, 1, 0, iters), tb = run<KIND>(d, 0, 1, iters), tab = run<KIND>(d, 1, 1, iters);
    printf("%-28s A alone (32 MFMAs/iter) %8.1f us   B alone (256 v_fma/iter) %8.1f us   both %8.1f us   max %8.1f  sum %8.1f  -> overlap %.0f %%\n",
           name, ta, tb, tab, ta > tb ? ta : tb, ta + tb, 100.0 * (ta + tb - tab) / (ta < tb ? ta : tb));
}

int main() {
    float *d; hipMalloc(&d, 64);
    report<0>("v_mfma_f32_32x32x2_f32", d);
    report<1>("v_mfma_i32_32x32x32_i8", d);
    report<2>("v_mfma_f32_32x32x16_bf16", d);
    return 0;
}
