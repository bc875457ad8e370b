// Probe for a dynamically scheduled env step (round 4): (1) what does ONE agent-scope fetch-add per wavefront on a shared counter
// cost when 2048 one-wavefront workgroups issue it in their first instructions (latency seen by a wave, launch time), one counter
// for the chip against one per XCD; (2) do the XCDs stream stores at the same rate -- 2048 wavefronts each write 25 600 contiguous
// bytes (the two groups x two viewers of a cramped_room wavefront) with `sc1 nt` 16-byte stores into a rotating 64-slot destination,
// end time per XCC id, with the block -> address mapping as the env kernel has it (block b writes chunk b) and rotated by one
// (block b writes chunk b+1: an address effect moves to the other XCDs, a physical one stays).
// Build: hipcc --offload-arch=gfx950 -O3 -o ticket_atomic ticket_atomic.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF; }

// mode 0: no atomic; 1: one counter; 2: one counter per XCC (256 bytes apart)
__global__ void k_atomic(unsigned *ctr, unsigned long long *trace, int mode) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned v = 0;
    if (mode && threadIdx.x == 0) {
        unsigned *p = ctr + (mode == 2 ? xcc_id() * 64 : 0);
        v = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    v = __builtin_amdgcn_readfirstlane(v);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime() + (v & 0);
    if (threadIdx.x == 0) { trace[blockIdx.x * 4] = t0; trace[blockIdx.x * 4 + 1] = t1; trace[blockIdx.x * 4 + 2] = v; trace[blockIdx.x * 4 + 3] = xcc_id(); }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_stream(char *dst, int rot, int bytes_per_wave, unsigned long long *trace) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int chunk = (blockIdx.x + rot) % gridDim.x;
    char *p = dst + (size_t)chunk * bytes_per_wave + threadIdx.x * 16;
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int o = 0; o + 1024 <= bytes_per_wave; o += 1024)
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p + o), "v"(v) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && trace) { trace[blockIdx.x * 4] = t0; trace[blockIdx.x * 4 + 1] = t1; trace[blockIdx.x * 4 + 3] = xcc_id(); }
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0 : v[v.size() / 2]; }

int main() {
    const int grid = 2048;
    unsigned *ctr; unsigned long long *trace;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&trace, grid * 32));
    std::vector<unsigned long long> h(grid * 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(ctr, 0, 4096));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(64), 0, 0, ctr, trace, mode);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(64), 0, 0, ctr, trace, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), trace, grid * 32, hipMemcpyDeviceToHost));
        std::vector<double> lat; double first = 1e30, last = 0;
        for (int b = 0; b < grid; ++b) { lat.push_back((h[b * 4 + 1] - h[b * 4]) / 100.0); first = std::min(first, (double)h[b * 4]); last = std::max(last, (double)h[b * 4 + 1]); }
        std::sort(lat.begin(), lat.end());
        printf("atomic mode %d (%s): launch %.2f us; per-wave latency min %.2f median %.2f p90 %.2f max %.2f us; first start -> last return %.2f us\n", mode,
               mode == 0 ? "none" : mode == 1 ? "one counter" : "one counter per XCC", ms * 1000 / 200, lat[0], lat[grid / 2], lat[grid * 9 / 10], lat[grid - 1], (last - first) / 100.0);
    }
    // ---- store streams by XCC
    const int bpw = 25600, slots = 64;
    char *dst; CK(hipMalloc(&dst, (size_t)slots * grid * bpw));
    for (int rot = 0; rot < 2; ++rot) {
        for (int i = 0; i < slots; ++i) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(64), 0, 0, dst + (size_t)i * grid * bpw, rot, bpw, (unsigned long long *)nullptr);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 4; ++r)
            for (int i = 0; i < slots; ++i) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(64), 0, 0, dst + (size_t)i * grid * bpw, rot, bpw, (unsigned long long *)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        hipLaunchKernelGGL(k_stream, dim3(grid), dim3(64), 0, 0, dst + (size_t)7 * grid * bpw, rot, bpw, trace);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), trace, grid * 32, hipMemcpyDeviceToHost));
        double first = 1e30;
        for (int b = 0; b < grid; ++b) first = std::min(first, (double)h[b * 4]);
        printf("stream rot %d: %.2f us per launch of %.1f MB; end (us after first start) by XCC:", rot, ms * 1000 / (4 * slots), grid * (double)bpw / 1e6);
        for (int x = 0; x < 8; ++x) {
            std::vector<double> e; double mx = 0;
            for (int b = 0; b < grid; ++b) if ((int)h[b * 4 + 3] == x) { e.push_back((h[b * 4 + 1] - first) / 100.0); mx = std::max(mx, e.back()); }
            printf("  %d: med %.2f max %.2f", x, med(e), mx);
        }
        printf("\n");
    }
    return 0;
}
