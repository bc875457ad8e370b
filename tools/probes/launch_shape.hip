// Probe: what does a launch of 2048 wavefronts cost as 2048 x 64, 1024 x 128, 512 x 256, 256 x 512 threads (workgroups x size), back to
// back on one stream -- empty kernels, and kernels whose every wavefront loads 1 KB and stores 1 KB (so that the end-of-kernel
// write-back has something to do).  The env step kernel runs one wavefront per workgroup; this asks whether
// the dispatcher charges for that.   Build: hipcc --offload-arch=gfx950 -O3 -o launch_shape launch_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_touch(const u32x4 *in, u32x4 *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v = in[i];
    v.x += 1;
    out[i] = v;
}
// LDS per workgroup as the env kernel has it (~14 KB per wavefront): does the allocation change the dispatch rate?
__global__ void k_lds(int *p) { extern __shared__ int lds[]; if (p && threadIdx.x == 9999) p[0] = lds[0]; }
int main() {
    int *p; u32x4 *a, *b;
    CK(hipMalloc(&p, 64)); CK(hipMalloc(&a, 2048 * 64 * 16)); CK(hipMalloc(&b, 2048 * 64 * 16));
    CK(hipMemset(a, 0, 2048 * 64 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[5][2] = {{2048, 64}, {1024, 128}, {512, 256}, {256, 512}, {128, 1024}};
    for (int kind = 0; kind < 3; ++kind)
        for (auto &sh : shapes) {
            auto go = [&]() {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), 0, 0, p);
                else if (kind == 1) hipLaunchKernelGGL(k_touch, dim3(sh[0]), dim3(sh[1]), 0, 0, a, b);
                else hipLaunchKernelGGL(k_lds, dim3(sh[0]), dim3(sh[1]), 14 * 1024 * (sh[1] / 64), 0, p);
            };
            if (kind == 2 && sh[1] > 256) continue;                  // 64 KB of LDS per workgroup at most
            for (int i = 0; i < 50; ++i) go();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 1000; ++i) go();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s %4d x %4d: %.2f us per launch\n", kind == 0 ? "empty" : kind == 1 ? "1 KB in + out per wave" : "empty, 14 KB LDS per wave", sh[0], sh[1], ms);
        }
    // how the cost of an empty launch grows with its wavefronts (one-wavefront workgroups)
    for (int waves : {8, 256, 512, 1024, 2048, 4096, 8192}) {
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(waves), dim3(64), 0, 0, p);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(waves), dim3(64), 0, 0, p);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty %5d x   64: %.2f us per launch\n", waves, ms);
    }
    return 0;
}
