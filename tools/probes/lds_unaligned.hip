// lds_unaligned.hip (round 5) -- do LDS dword / 16-bit / 128-bit accesses work at byte addresses that are not multiples of their size on gfx950
// (SH_MEM_CONFIG.alignment_mode = unaligned, which the Linux compute driver sets for gfx9), and what do they cost?
// The N-player step kernel stages tiles of C = 5P + 10 = 15 / 25 / 30 bytes back to back in LDS (tile t at byte t * C) so that the
// flush can read the viewers' planes as contiguous 16-byte chunks: every tile is then a handful of dword writes at an arbitrary
// byte alignment plus a 16-bit / 8-bit tail.
//   hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t __attribute__((aligned(1))) u32_u;
typedef uint16_t __attribute__((aligned(1))) u16_u;

template <int C>
__global__ void stage_kernel(uint8_t *out, unsigned long long *cycles, int reps) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        uint8_t *p = lds + lane * C;
        // tile bytes: byte i of tile `lane` = (lane * 7 + i * 3 + r) & 0xff
        auto byte = [&](int i) { return static_cast<uint32_t>((lane * 7 + i * 3 + r) & 0xff); };
        int i = 0;
#pragma unroll
        for (; i + 4 <= C; i += 4) *reinterpret_cast<u32_u *>(p + i) = byte(i) | (byte(i + 1) << 8) | (byte(i + 2) << 16) | (byte(i + 3) << 24);
        if (C - i >= 2) { *reinterpret_cast<u16_u *>(p + i) = static_cast<uint16_t>(byte(i) | (byte(i + 1) << 8)); i += 2; }
        if (C - i >= 1) p[i] = static_cast<uint8_t>(byte(i));
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // flush: 16 bytes per lane, aligned
    for (int c = lane; c * 16 < 64 * C; c += 64) reinterpret_cast<uint4 *>(out)[c] = reinterpret_cast<const uint4 *>(lds)[c];
    if (lane == 0) cycles[0] = t1 - t0;
}

template <int C>
int run() {
    uint8_t *d_out; unsigned long long *d_cyc;
    const int bytes = (64 * C + 15) / 16 * 16, reps = 1000;
    hipMalloc(&d_out, bytes); hipMalloc(&d_cyc, 8);
    hipMemset(d_out, 0xEE, bytes);
    hipLaunchKernelGGL(stage_kernel<C>, dim3(1), dim3(64), bytes, 0, d_out, d_cyc, reps);
    std::vector<uint8_t> h(bytes); unsigned long long cyc = 0;
    hipMemcpy(h.data(), d_out, bytes, hipMemcpyDeviceToHost); hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t)
        for (int i = 0; i < C; ++i) bad += h[t * C + i] != static_cast<uint8_t>((t * 7 + i * 3 + reps - 1) & 0xff);
    printf("C = %2d: %d mismatching bytes of %d; %.1f memtime ticks (100 MHz) per staged pass of 64 tiles\n", C, bad, 64 * C, double(cyc) / reps);
    hipFree(d_out); hipFree(d_cyc);
    return bad;
}

int main() {
    int bad = run<15>() + run<25>() + run<30>() + run<20>() + run<32>();
    printf(bad ? "UNALIGNED LDS ACCESS DOES NOT WORK HERE\n" : "unaligned LDS stores work\n");
    return bad != 0;
}
