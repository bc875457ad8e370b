// Probe (round 4): does the ORDER in which the step kernel's plane bytes reach memory matter?  52.4 MB per launch into fresh destinations
// (64 in rotation, nothing fits the Infinity Cache), 2048 one-wavefront workgroups, `sc1 nt` 16-byte stores, back to back launches.
//   sweep     every store instruction of the chip writes one contiguous 2 MB window (wave w, iteration i -> KB i * 2048 + w): the
//             pattern of tools/probes/store_ceiling.hip
//   runs      what the step kernel does: two viewer planes (26.2 MB apart), wave w owns the 6400-byte runs of groups w and w + 2048
//             in both planes and writes each as flushes of 2560 + 3840 bytes, viewer 0 then viewer 1
//   runs2     the same with the wave's two groups adjacent (groups 2w, 2w + 1: 12 800-byte runs)
//   rot       runs, every wave starting its run at a different 1280-byte phase (w mod 5) and wrapping around
//   oneplane  runs with both viewers' bytes in one plane (wave w: one 25 600-byte run)
// Build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void st(char *p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 0" : : "v"(p), "v"(v) : "memory"); }

constexpr int kRun = 6400, kWaves = 2048, kGroups = 4096;
constexpr long long kPlane = (long long)kGroups * kRun;            // 26 214 400 bytes per viewer

// bytes [off, off + n) of a run at `base`, 16 bytes per lane, whole 1 KB instructions and a partial last one
__device__ __forceinline__ void put(char *base, int off, int n, int lane, u32x4 v) {
    for (int o = lane * 16; o < n; o += 1024) st(base + off + o, v);
}

template <int MODE> __global__ __launch_bounds__(64) void k(char *dst) {
    const int lane = threadIdx.x, w = blockIdx.x;
    const u32x4 v = {1u, 2u, 3u, (unsigned)w};
    if (MODE == 0) {                                               // sweep
        const long long total = 2 * kPlane / 16;
        for (long long i = w * 64ll + lane; i < total; i += kWaves * 64ll) st(dst + i * 16, v);
        return;
    }
    for (int gi = 0; gi < 2; ++gi) {
        const int g = MODE == 2 ? 2 * w + gi : w + gi * kWaves;
        if (MODE == 4) {                                           // one plane: the wave's 25 600 bytes contiguous
            char *base = dst + (long long)w * 4 * kRun + (long long)gi * 2 * kRun;
            put(base, 0, 2560, lane, v); put(base, 2560, 2560, lane, v); put(base, 5120, 3840, lane, v); put(base, 8960, 3840, lane, v);
            continue;
        }
        char *b0 = dst + (long long)g * kRun, *b1 = b0 + kPlane;
        if (MODE == 3) {                                           // rotated start: five 1280-byte pieces, starting at piece w mod 5
            const int ph = w % 5;
            for (int piece = 0; piece < 5; ++piece) {
                const int off = ((piece + ph) % 5) * 1280;
                put(b0, off, 1280, lane, v); put(b1, off, 1280, lane, v);
            }
            continue;
        }
        put(b0, 0, 2560, lane, v); put(b1, 0, 2560, lane, v);
        put(b0, 2560, 3840, lane, v); put(b1, 2560, 3840, lane, v);
    }
}

template <int MODE> int run(const char *name, char *ring, int slots) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long stride = 2 * kPlane;
    for (int i = 0; i < slots; ++i) hipLaunchKernelGGL((k<MODE>), dim3(kWaves), dim3(64), 0, 0, ring + (i % slots) * stride);
    CK(hipEventRecord(e0));
    const int K = 640;
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL((k<MODE>), dim3(kWaves), dim3(64), 0, 0, ring + (i % slots) * stride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s %6.2f us per launch, %6.0f GB/s\n", name, ms * 1e3 / K, 2.0 * kPlane / (ms * 1e3 / K) / 1e3);
    return 0;
}

int main() {
    const int slots = 64;
    char *ring;
    CK(hipMalloc(&ring, 2 * kPlane * slots));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<0>("sweep", ring, slots) || run<1>("runs", ring, slots) || run<2>("runs2", ring, slots) || run<3>("rot", ring, slots) || run<4>("oneplane", ring, slots)) return 1;
    }
    return 0;
}
