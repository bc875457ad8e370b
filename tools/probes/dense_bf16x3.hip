// Probe: one conv position + one fc1 position of the policy kernel, registers only (no LDS), two ways:
//   A  (the kernel today)  24 x v_mfma_i32_32x32x32_i8  +  32 x v_mfma_f32_32x32x2_f32, conv accumulators -> f32 operand (7 VALU / register)
//   B  (bf16 x 3 split)    24 x v_mfma_i32_32x32x32_i8  +  24 x v_mfma_f32_32x32x16_bf16 (6 product terms x 2 k-halves x 2 output halves),
//                          conv accumulators -> f32 -> three exact bf16 pieces (truncation split), packed in pairs
// Cycles per position with one and two wavefronts per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
union B8 { v8bf v; unsigned u[4]; };

__device__ __forceinline__ float combine(const v16i (&acc)[4], int r, float sc, float bi) {
    float t = static_cast<float>(acc[0][r] * 128 + acc[1][r]);
    t = __builtin_fmaf(t, 16384.0f, static_cast<float>(acc[2][r] * 128 + acc[3][r]));
    return fmaxf(__builtin_fmaf(t, sc, bi), 0.f);
}

template <int MODE>   // 0: A, 1: B, 2: B without the split (MFMAs + combine only), 3: B MFMAs only
__global__ __launch_bounds__(512) void probe(unsigned long long *out, float seed, const int *ain, int iters) {
    v16i acc[4];
    v4i wq[6][4], bfr[6];
    for (int k = 0; k < 6; ++k) {
        for (int d = 0; d < 4; ++d) wq[k][d] = v4i{ain[k * 4 + d], ain[k + d + 1], ain[k + 2], ain[d + 3]};
        bfr[k] = v4i{ain[k], ain[k + 1], ain[k + 2], ain[k + 3]};
    }
    const float sc = seed * 1e-6f, bi = seed * 0.5f;
    v16f h0, h1, x;
    for (int r = 0; r < 16; ++r) { h0[r] = 0; h1[r] = 0; x[r] = seed + r; }
    float w[32];
    for (int i = 0; i < 32; ++i) w[i] = seed * i + threadIdx.x;
    B8 wa[3][2][2];                        // weight pieces [piece][k half][out half]
    for (int p = 0; p < 3; ++p) for (int m = 0; m < 2; ++m) for (int o = 0; o < 2; ++o) for (int i = 0; i < 4; ++i) wa[p][m][o].u[i] = ain[p + m + o + i] * 0x3f803f80u;
    B8 xb[3][2];                           // operand pieces of the current position [piece][k half]
    for (int p = 0; p < 3; ++p) for (int m = 0; m < 2; ++m) for (int i = 0; i < 4; ++i) xb[p][m].u[i] = 0x3f803f80u;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][r] = 0;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wq[ks][d], bfr[ks], acc[d], 0, 0, 0);
        bfr[it % 6].x += 1;
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * s], x[s], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2 * s + 1], x[s], h1, 0, 0, 0);
                x[s] = combine(acc, s, sc, bi);
            }
        } else {
            B8 nb[3][2];
            // terms in issue order: (weight piece, operand piece)
            constexpr int TW[6] = {0, 0, 1, 0, 1, 2}, TX[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[TW[t]][m][0].v, xb[TX[t]][m].v, h0, 0, 0, 0);
                    h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[TW[t]][m][1].v, xb[TX[t]][m].v, h1, 0, 0, 0);
                    if (MODE == 3) continue;
                    // VALU in the shadow: pairs of registers -> one packed dword per piece; 8 pairs over the 12 slots
                    const int j = m * 6 + t;                 // 0..11
                    if (j < 8) {
                        const float x0 = combine(acc, 2 * j, sc, bi), x1 = combine(acc, 2 * j + 1, sc, bi);
                        const int mm = j >> 2, dw = j & 3;
                        if (MODE == 2) { nb[0][mm].u[dw] = __float_as_uint(x0); nb[1][mm].u[dw] = __float_as_uint(x1); nb[2][mm].u[dw] = 0; continue; }
                        const v2f xv = {x0, x1};
                        const v2f hv = {__uint_as_float(__float_as_uint(x0) & 0xffff0000u), __uint_as_float(__float_as_uint(x1) & 0xffff0000u)};
                        const v2f rv = xv - hv;
                        const v2f mv = {__uint_as_float(__float_as_uint(rv.x) & 0xffff0000u), __uint_as_float(__float_as_uint(rv.y) & 0xffff0000u)};
                        const v2f qv = rv - mv;
                        nb[0][mm].u[dw] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
                        nb[1][mm].u[dw] = __builtin_amdgcn_perm(__float_as_uint(rv.y), __float_as_uint(rv.x), 0x07060302u);
                        nb[2][mm].u[dw] = __builtin_amdgcn_perm(__float_as_uint(qv.y), __float_as_uint(qv.x), 0x07060302u);
                    }
                }
            if (MODE != 3)
#pragma unroll
                for (int p = 0; p < 3; ++p) { xb[p][0] = nb[p][0]; xb[p][1] = nb[p][1]; }
        }
    }
    unsigned long long t1 = clock64();
    float a = 0;
    for (int r = 0; r < 16; ++r) a += h0[r] + h1[r] + x[r] + acc[0][r];
    if (a == 12345.f) out[100000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char *name, unsigned long long *d, int *ain, int threads) {
    const int blocks = 256, iters = 512;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f, ain, iters);
    hipDeviceSynchronize();
    static unsigned long long h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < threads / 64; ++w) { sum += h[b * 8 + w]; ++n; }
    const double per_pos = sum / n / iters;
    printf("%-72s %d waves/SIMD: %7.0f cycles per position per wave, %7.0f per SIMD-position\n", name, threads / 256, per_pos, per_pos / (threads / 256) );
}

int main() {
    unsigned long long *d; hipMalloc(&d, 8 * 100008); int *ain; hipMalloc(&ain, 4 * 64); hipMemset(ain, 1, 4 * 64);
    for (int threads : {256, 512}) {
        run<0>("A: i8 conv + 32 f32 MFMAs + combine", d, ain, threads);
        run<1>("B: i8 conv + 24 bf16 MFMAs + combine + exact 3-piece split", d, ain, threads);
        run<2>("B without the split (combine only)", d, ain, threads);
        run<3>("B MFMAs only", d, ain, threads);
    }
    return 0;
}
