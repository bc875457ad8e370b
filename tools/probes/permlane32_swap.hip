// Probe (round 6): v_permlane32_swap (gfx950) as the exchange between the two 32-lane halves of a wavefront -- __shfl_xor(x, 32) as ONE
// vector instruction + a select instead of ds_bpermute_b32 (an LDS round trip, ~100 cycles of latency on a dependent chain: the policy
// kernel epilogue has four in a row).  Build: hipcc --offload-arch=gfx950 -O3 -o permlane32_swap permlane32_swap.hip ; prints "bad 0".
#include <hip/hip_runtime.h>
__device__ __forceinline__ float xor32(float x) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const v2u r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r.x : r.y);
}
__global__ void k(float *o) { float x = threadIdx.x * 1.5f; o[threadIdx.x] = xor32(x) ; }
int main(){ float *d; hipMalloc(&d, 256); k<<<1,64>>>(d); float h[64]; hipMemcpy(h,d,256,hipMemcpyDeviceToHost); int bad=0; for(int i=0;i<64;++i) if (h[i]!=(i^32)*1.5f) ++bad; printf("bad %d h0 %f h33 %f\n", bad, h[0], h[33]); return bad; }
