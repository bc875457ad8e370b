// Probe: what one launch that only streams the step kernel's bytes costs on this part -- 59.5 MB of 16-byte stores
// (observation planes + state + reward/done of 65 536 cramped_room kitchens) and 6.8 MB of loads, nothing else,
// launched back to back on one stream like the env steps.  The practical floor for a one-launch-per-step kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_ceiling store_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <initializer_list>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, bool READ>
__global__ __launch_bounds__(64) void stream(u32x4 *dst, const u32x4 *src, long long n_store, long long n_load) {
    const long long tid = blockIdx.x * 64ll + threadIdx.x, nthreads = gridDim.x * 64ll;
    u32x4 v = {1u, 2u, 3u, (unsigned)tid};
    if (READ)
        for (long long i = tid; i < n_load; i += nthreads) { u32x4 t = src[i]; v.x ^= t.x; v.y += t.w; }
    for (long long i = tid; i < n_store; i += nthreads) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

template <bool NT, bool READ> void run(const char *name, u32x4 *dst, u32x4 *src, long long store_bytes, long long load_bytes, int waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((stream<NT, READ>), dim3(waves), dim3(64), 0, 0, dst, src, store_bytes / 16, load_bytes / 16);
    hipEventRecord(e0);
    const int K = 1000;
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL((stream<NT, READ>), dim3(waves), dim3(64), 0, 0, dst, src, store_bytes / 16, load_bytes / 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / K;
    printf("%-46s %5d waves: %6.2f us per launch, %6.0f GB/s\n", name, waves, us, (store_bytes + (READ ? load_bytes : 0)) / us / 1e3);
}

// The same launches with a FRESH destination each (64 buffers in rotation, 3.8 GB: nothing is absorbed by the 256 MB Infinity
// Cache): the floor of the time-major ("HBM proper") step.  SM: 0 plain, 1 nt, 2 sc1, 4 sc1 nt (inline asm like the step kernel's).
template <int SM>
__global__ __launch_bounds__(64) void stream_sm(u32x4 *dst, const u32x4 *src, long long n_store, long long n_load) {
    const long long tid = blockIdx.x * 64ll + threadIdx.x, nthreads = gridDim.x * 64ll;
    u32x4 v = {1u, 2u, 3u, (unsigned)tid};
    for (long long i = tid; i < n_load; i += nthreads) { u32x4 t = src[i]; v.x ^= t.x; v.y += t.w; }
    for (long long i = tid; i < n_store; i += nthreads) {
        u32x4 *p = dst + i;
        if (SM == 1) __builtin_nontemporal_store(v, p);
        else if (SM == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
        else if (SM == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
        else *p = v;
    }
}

template <int SM> void run_fresh(const char *name, u32x4 *ring, u32x4 *src, long long store_bytes, long long load_bytes, int waves, int slots) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long stride = (store_bytes / 16 + 255) / 256 * 256;
    for (int i = 0; i < slots; ++i) hipLaunchKernelGGL((stream_sm<SM>), dim3(waves), dim3(64), 0, 0, ring + (i % slots) * stride, src, store_bytes / 16, load_bytes / 16);
    hipEventRecord(e0);
    const int K = 640;
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL((stream_sm<SM>), dim3(waves), dim3(64), 0, 0, ring + (i % slots) * stride, src, store_bytes / 16, load_bytes / 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / K;
    printf("%-46s %5d waves: %6.2f us per launch, %6.0f GB/s\n", name, waves, us, (store_bytes + load_bytes) / us / 1e3);
}

int main() {
    const long long store_bytes = 59506688, load_bytes = 6815744;     // writes / reads of one step at 65 536 kitchens
    u32x4 *dst, *src; hipMalloc(&dst, store_bytes); hipMalloc(&src, load_bytes); hipMemset(src, 1, load_bytes);
    for (int waves : {2048, 4096, 8192, 65536}) {
        run<false, false>("stores only, in place", dst, src, store_bytes, load_bytes, waves);
        run<false, true>("loads then stores, in place", dst, src, store_bytes, load_bytes, waves);
        run<true, true>("loads then non-temporal stores, in place", dst, src, store_bytes, load_bytes, waves);
    }
    {
        const int slots = 64;
        const long long stride = (store_bytes / 16 + 255) / 256 * 256;
        u32x4 *ring = nullptr;
        if (hipMalloc(&ring, stride * 16 * slots) != hipSuccess) { printf("no memory for the fresh-destination ring\n"); return 0; }
        for (int waves : {1536, 2048, 3072, 4096}) {
            run_fresh<0>("loads then plain stores, fresh destination", ring, src, store_bytes, load_bytes, waves, slots);
            run_fresh<1>("loads then nt stores, fresh destination", ring, src, store_bytes, load_bytes, waves, slots);
            run_fresh<2>("loads then sc1 stores, fresh destination", ring, src, store_bytes, load_bytes, waves, slots);
            run_fresh<4>("loads then sc1 nt stores, fresh destination", ring, src, store_bytes, load_bytes, waves, slots);
        }
    }
    return 0;
}
