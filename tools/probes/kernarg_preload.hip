// Probe: kernel-argument preload (hipcc -mllvm -amdgpu-kernarg-preload-count=N): the leading scalar arguments of a kernel arrive in SGPRs
// with the wavefront instead of through s_load -- does the firmware of this GPU honour it, i.e. does a kernel built that way compute
// with the right values?  (A build for firmware without the feature falls into the compiler's compatibility prologue, which loads them
// the old way: correct either way; `.amdhsa_user_sgpr_kernarg_preload_length` in the ISA says whether the build asks for it.)
// Struct arguments are not preloaded.  The per-engine env kernels are built this way (overcooked_engine.hip, kJitPreloadOption).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -o kernarg_preload kernarg_preload.hip ; prints "13 13 (want 13)".
#include <hip/hip_runtime.h>
#include <cstdio>
struct Rest { int a, b, c, d; long long e; };
__global__ void k(const int *in, int *out, long long n, int add, Rest r) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add + r.a + (int)r.e;
}
int main() {
    int *in, *out; const int n = 1 << 20;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    hipMemset(in, 0, n * 4);
    Rest r{1, 2, 3, 4, 5};
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, in, out, (long long)n, 7, r);
    int h[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%d %d (want 13)\n", h[0], h[3]);
    return h[0] == 13 ? 0 : 1;
}
