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
