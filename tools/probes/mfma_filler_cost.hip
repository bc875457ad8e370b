// Probe (round 6): what ONE wavefront's vector instructions of the policy kernel's operand conversion cost beside its own MFMAs
// (v_mfma_f32_32x32x16_f16, one wavefront per SIMD): N copies of one instruction kind behind every MFMA, N = 0, 3, 4, 5, 6, 7.
// Up to five 4-cycle instructions hide in a 32-cycle MFMA gap; the sixth costs ~22 cycles (mfma_valu_overlap.hip) -- this probe finds
// which of the conversion's instructions count as more than one.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_filler_cost mfma_filler_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define REP0(S) ""
#define REP3(S) S S S
#define REP4(S) S S S S
#define REP5(S) S S S S S
#define REP6(S) S S S S S S
#define REP7(S) S S S S S S S
#define MF "v_mfma_f32_32x32x16_f16 %[a0], %[wa], %[wb], %[a0]\n"
#define MF1 "v_mfma_f32_32x32x16_f16 %[a1], %[wa], %[wb], %[a1]\n"
#define OPS : [a0] "+v"(a0), [a1] "+v"(a1), [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [ix] "+v"(ix), [iy] "+v"(iy) : [wa] "v"(wa), [wb] "v"(wb), [l] "v"(ldsaddr)

#define KERNEL(NAME, INS)                                                                                                        \
    template <int N> __global__ __launch_bounds__(256) void NAME(float *sink, long long *cyc, float seed, int iters) {            \
        v16f a0, a1;                                                                                                              \
        for (int r = 0; r < 16; ++r) { a0[r] = seed; a1[r] = seed; }                                                              \
        v4i wa = {(int)threadIdx.x, 1, 2, 3}, wb = {4, 5, 6, (int)threadIdx.x};                                                   \
        float x = seed, y = seed + 1.f, z = 0.f;                                                                                  \
        int ix = threadIdx.x, iy = 3;                                                                                             \
        unsigned ldsaddr = threadIdx.x * 8;                                                                                       \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                                        \
        for (int it = 0; it < iters; ++it) {                                                                                      \
            if (N == 0) asm volatile(MF REP0(INS) MF1 REP0(INS) OPS);                                                             \
            if (N == 3) asm volatile(MF REP3(INS) MF1 REP3(INS) OPS);                                                             \
            if (N == 4) asm volatile(MF REP4(INS) MF1 REP4(INS) OPS);                                                             \
            if (N == 5) asm volatile(MF REP5(INS) MF1 REP5(INS) OPS);                                                             \
            if (N == 6) asm volatile(MF REP6(INS) MF1 REP6(INS) OPS);                                                             \
            if (N == 7) asm volatile(MF REP7(INS) MF1 REP7(INS) OPS);                                                             \
        }                                                                                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                        \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                                        \
        float s = x + y + z + ix + iy;                                                                                            \
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];                                                                          \
        if (s == 12345.f) sink[0] = s;                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                                                \
    }

KERNEL(k_fma, "v_fma_f32 %[x], %[x], %[y], %[y]\n")
KERNEL(k_mul, "v_mul_f32 %[z], %[x], %[y]\n")
KERNEL(k_fmac, "v_fmac_f32 %[z], %[x], %[y]\n")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %[z], %[ix]\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %[iy], %[ix], 8, %[ix]\n")
KERNEL(k_med3, "v_med3_f32 %[z], %[x], 0, %[y]\n")
KERNEL(k_cvt_pk, "v_cvt_pk_f16_f32 %[z], %[x], %[y]\n")
KERNEL(k_fma_mix, "v_fma_mix_f32 %[z], %[x], -1.0, %[y] op_sel_hi:[1,0,0]\n")
KERNEL(k_mov, "v_mov_b32 %[z], %[x]\n")
KERNEL(k_nop, "s_nop 0\n")

template <int N, class K>
double run1(K kern, float *d, long long *c) {
    const int iters = 4000;
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, c, 1.0f, iters);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    return (double)h / (2.0 * iters);
}
#define SWEEP(NAME, LABEL) printf("%-18s cycles per MFMA with 0/3/4/5/6/7 of them behind it: %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", LABEL, \
    run1<0>(NAME<0>, d, c), run1<3>(NAME<3>, d, c), run1<4>(NAME<4>, d, c), run1<5>(NAME<5>, d, c), run1<6>(NAME<6>, d, c), run1<7>(NAME<7>, d, c)); fflush(stdout);

int main() {
    float *d; long long *c;
    hipMalloc(&d, 64); hipMalloc(&c, 64);
    SWEEP(k_fma, "v_fma_f32") SWEEP(k_mul, "v_mul_f32") SWEEP(k_fmac, "v_fmac_f32") SWEEP(k_cvt_f32_i32, "v_cvt_f32_i32") SWEEP(k_lshl_add, "v_lshl_add_u32")
    SWEEP(k_med3, "v_med3_f32") SWEEP(k_cvt_pk, "v_cvt_pk_f16_f32") SWEEP(k_fma_mix, "v_fma_mix_f32") SWEEP(k_mov, "v_mov_b32") SWEEP(k_nop, "s_nop 0")
    return 0;
}
