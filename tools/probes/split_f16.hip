// Probe: the two-way f16 split of the policy kernel (v_cvt_pk_f16_f32, v_fma_mix_f32 remainders, v_cvt_pk_f16_f32) against a host
// restatement (round to nearest even, exact remainder) on 2^24 random pairs plus edge cases; and the three-term product
// hi*hi + hi*lo + lo*hi against the f32 product.   hipcc --offload-arch=gfx950 -O3 -o split_f16 split_f16.hip && ./split_f16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../diverse-conventions_amd/csrc/policy_pack.h"
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, int &hi, int &lo) {
    const v2f xv = {x0, x1};
    const v2h h = __builtin_convertvector(xv, v2h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    const v2f rv = {r0, r1};
    const v2h l = __builtin_convertvector(rv, v2h);
    hi = __builtin_bit_cast(int, h);
    lo = __builtin_bit_cast(int, l);
}
__global__ void k(const float *x, int *hi, int *lo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) split_pair(x[2 * i], x[2 * i + 1], hi[i], lo[i]);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> x(2 * n);
    uint32_t s = 12345;
    for (int i = 0; i < 2 * n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float mant = 1.0f + (s >> 9) * (1.0f / 8388608.0f);
        s = s * 1664525u + 1013904223u;
        x[i] = std::ldexp(mant, (int)(s >> 27) - 20);                  // 2^-20 .. 2^12
    }
    x[0] = 0.f; x[1] = 65504.f; x[2] = 1.f; x[3] = 0.333333343f; x[4] = 6.1e-5f; x[5] = 5.9e-8f; x[6] = 4095.999f; x[7] = 2048.5f;
    float *dx; int *dh, *dl;
    hipMalloc(&dx, 2 * n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dx, x.data(), 2 * n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dh, dl, n);
    std::vector<int> hi(n), lo(n);
    hipMemcpy(hi.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lo.data(), dl, n * 4, hipMemcpyDeviceToHost);
    long long bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 2; ++j) {
            const float v = x[2 * i + j];
            const uint16_t h = (uint16_t)((uint32_t)hi[i] >> (16 * j)), l = (uint16_t)((uint32_t)lo[i] >> (16 * j));
            const uint16_t wh = oc_f16_rne(v), wl = oc_f16_rne(v - oc_f16_to_f32(wh));
            if (h != wh || l != wl) { if (bad < 5) printf("x=%g got %04x %04x want %04x %04x\n", v, h, l, wh, wl); ++bad; }
            const double err = std::fabs((double)oc_f16_to_f32(h) + (double)oc_f16_to_f32(l) - (double)v) / (v != 0 ? std::fabs((double)v) : 1);
            if (v >= 1.0f / 64 && err > worst) worst = err;
        }
    printf("pairs %d mismatches vs host RNE split: %lld; worst relative error of hi+lo for x >= 1/64: %.3g (2^-24 = %.3g)\n", n, bad, worst, std::ldexp(1.0, -24));
    return bad != 0;
}
