// lds_dma_align.hip -- does global_load_lds_dwordx4 accept a global address that is only dword-aligned, and an LDS
// destination that is only dword-aligned?  (Gathered row staging of the policy kernel: a row of a 9x5 grid is 900 bytes,
// so row r of the observation tensor starts 4-byte aligned only.)  Also: a scalar-base + lane-offset form with the
// instruction's immediate offset (added to BOTH addresses).
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_align lds_dma_align.hip && ./lds_dma_align
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int SIZE>
__global__ void k(const uint8_t *src, int src_off, int lds_off, uint8_t *out) {
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds_raw);
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xEE;
    __syncthreads();
    const uint8_t *p = src + src_off + SIZE * threadIdx.x;
    if (SIZE == 16) __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + lds_off), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + lds_off), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<uint8_t> h(8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d, *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int size : {16, 4})
        for (int so : {0, 4, 8, 12, 900})
            for (int lo : {0, 4, 8, 16, 900 % 16 + 16}) {
                hipMemset(o, 0, 4096);
                if (size == 16) k<16><<<1, 64, 4096>>>(d, so, lo, o); else k<4><<<1, 64, 4096>>>(d, so, lo, o);
                hipError_t e = hipDeviceSynchronize();
                std::vector<uint8_t> r(4096);
                hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
                int bad = 0;
                for (int i = 0; i < 4096; ++i) {
                    const int j = i - lo;
                    const uint8_t want = (j >= 0 && j < 64 * size) ? h[so + j] : 0xEE;
                    bad += r[i] != want;
                }
                printf("size %2d  global +%3d  lds +%2d : %s (%d bytes differ) %s\n", size, so, lo, bad ? "MISMATCH" : "ok", bad, e == hipSuccess ? "" : hipGetErrorString(e));
                bad_total += bad != 0;
            }
    return bad_total ? 1 : 0;
}
