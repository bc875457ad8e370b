// Probe (round 6): do a wavefront's OWN vector instructions run under its MFMAs -- and does it matter whether the MFMA's accumulator
// lives in ArchVGPRs (-amdgpu-mfma-vgpr-form, what policy_ops.hip uses so that the VALU can read conv results directly) or in AccVGPRs?
// One wavefront per SIMD (256-thread workgroups, one per CU), a loop of [1 MFMA + F independent single-issue vector instructions],
// F = 0 .. 6, for v_mfma_f32_32x32x16_f16 and v_mfma_i32_32x32x32_i8, accumulators in "v" or "a" registers (inline asm pins the class).
// Prints cycles per MFMA (s_memtime).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define FILL1 "v_fma_f32 %[x0], %[x0], %[c], %[c]\n"
#define FILL2 FILL1 "v_fma_f32 %[x1], %[x1], %[c], %[c]\n"
#define FILL3 FILL2 "v_fma_f32 %[x2], %[x2], %[c], %[c]\n"
#define FILL4 FILL3 "v_fma_f32 %[x3], %[x3], %[c], %[c]\n"
#define FILL5 FILL4 "v_fma_f32 %[x4], %[x4], %[c], %[c]\n"
#define FILL6 FILL5 "v_fma_f32 %[x5], %[x5], %[c], %[c]\n"
#define FILL0 ""

// KIND 0: f16 32x32x16, 1: i8 32x32x32.  ACC 0: VGPR accumulators, 1: AGPR accumulators.
#define BODY(MF, CLS, FILL)                                                                                                        \
    asm volatile(MF " %[a0], %[wa], %[wb], %[a0]\n" FILL MF " %[a1], %[wa], %[wb], %[a1]\n" FILL                                   \
                 MF " %[a2], %[wa], %[wb], %[a2]\n" FILL MF " %[a3], %[wa], %[wb], %[a3]\n" FILL                                   \
                 : [a0] "+" CLS(a0), [a1] "+" CLS(a1), [a2] "+" CLS(a2), [a3] "+" CLS(a3), [x0] "+v"(x[0]), [x1] "+v"(x[1]),       \
                   [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5])                                              \
                 : [wa] "v"(wa), [wb] "v"(wb), [c] "v"(c));

template <int KIND, int ACC, int F>
__global__ __launch_bounds__(256) void probe(float *sink, long long *cyc, float seed, int iters) {
    v16f a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = seed; a1[r] = seed; a2[r] = seed; a3[r] = seed; }
    v4i wa = {(int)threadIdx.x, 1, 2, 3}, wb = {4, 5, 6, (int)threadIdx.x};
    float x[6];
    for (int i = 0; i < 6; ++i) x[i] = seed + i;
    const float c = 1.0001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define RUN(MF, CLS)                                                                     \
        if (F == 0) { BODY(MF, CLS, FILL0) } else if (F == 1) { BODY(MF, CLS, FILL1) }   \
        else if (F == 2) { BODY(MF, CLS, FILL2) } else if (F == 3) { BODY(MF, CLS, FILL3) } \
        else if (F == 4) { BODY(MF, CLS, FILL4) } else if (F == 5) { BODY(MF, CLS, FILL5) } else { BODY(MF, CLS, FILL6) }
        if (KIND == 0 && ACC == 0) { RUN("v_mfma_f32_32x32x16_f16", "v") }
        if (KIND == 0 && ACC == 1) { RUN("v_mfma_f32_32x32x16_f16", "a") }
        if (KIND == 1 && ACC == 0) { RUN("v_mfma_i32_32x32x32_i8", "v") }
        if (KIND == 1 && ACC == 1) { RUN("v_mfma_i32_32x32x32_i8", "a") }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    for (int i = 0; i < 6; ++i) s += x[i];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int ACC, int F>
void run(float *d, long long *c) {
    const int iters = 4000;
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((probe<KIND, ACC, F>), dim3(256), dim3(256), 0, 0, d, c, 1.0f, iters);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-26s acc in %s  %d fillers per MFMA: %6.1f cycles per MFMA\n", KIND ? "v_mfma_i32_32x32x32_i8" : "v_mfma_f32_32x32x16_f16", ACC ? "AGPR" : "VGPR", F,
           (double)h / (4.0 * iters));
}

template <int KIND, int ACC>
void sweep(float *d, long long *c) {
    run<KIND, ACC, 0>(d, c); run<KIND, ACC, 1>(d, c); run<KIND, ACC, 2>(d, c); run<KIND, ACC, 3>(d, c);
    run<KIND, ACC, 4>(d, c); run<KIND, ACC, 5>(d, c); run<KIND, ACC, 6>(d, c);
}

int main() {
    float *d; long long *c;
    hipMalloc(&d, 64); hipMalloc(&c, 64);
    sweep<0, 0>(d, c); sweep<0, 1>(d, c); sweep<1, 0>(d, c); sweep<1, 1>(d, c);
    return 0;
}
