// Probe (round 6, follow-up of mfma_valu_overlap.hip): WHICH property of the policy kernel's vector work stops it from running under
// the wavefront's own MFMAs?  One wavefront per SIMD; a loop of 4 x [MFMA + 4 vector instructions], everything in explicitly named
// registers (one asm block), four accumulator tuples v[64:79] v[80:95] v[96:111] v[112:127] written in turn.  Variants:
//   0 fillers v_fma_f32 on private registers (the baseline of the first probe)
//   1 fillers v_cvt_f32_i32 READING accumulator registers (of the tuple written two MFMAs ago), writing private registers
//   2 fillers v_lshl_add_u32 with two accumulator sources (different tuples), writing private registers
//   3 as 1, but the MFMA's A and B operands come from AccVGPRs
//   4 fillers v_cvt_f32_i32 reading accumulator registers AND writing into a fifth tuple's registers that later MFMAs use as B operand
//   5 no fillers
// Prints cycles per MFMA.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap2 mfma_valu_overlap2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
  "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
  "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
  "a0","a1","a2","a3","a4","a5","a6","a7"

#define M(D, A, B) "v_mfma_f32_32x32x16_f16 v[" D "], " A ", " B ", v[" D "]\n"
#define VA "v[0:3]"
#define VB "v[4:7]"
#define AA "a[0:3]"
#define AB "a[4:7]"

template <int V>
__global__ __launch_bounds__(256) void probe(long long *cyc, int iters) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) asm volatile(
            M("64:79", VA, VB) "v_fma_f32 v8, v8, v20, v20\n v_fma_f32 v9, v9, v20, v20\n v_fma_f32 v10, v10, v20, v20\n v_fma_f32 v11, v11, v20, v20\n"
            M("80:95", VA, VB) "v_fma_f32 v12, v12, v20, v20\n v_fma_f32 v13, v13, v20, v20\n v_fma_f32 v14, v14, v20, v20\n v_fma_f32 v15, v15, v20, v20\n"
            M("96:111", VA, VB) "v_fma_f32 v8, v8, v20, v20\n v_fma_f32 v9, v9, v20, v20\n v_fma_f32 v10, v10, v20, v20\n v_fma_f32 v11, v11, v20, v20\n"
            M("112:127", VA, VB) "v_fma_f32 v12, v12, v20, v20\n v_fma_f32 v13, v13, v20, v20\n v_fma_f32 v14, v14, v20, v20\n v_fma_f32 v15, v15, v20, v20\n" ::: CLOB);
        if (V == 1) asm volatile(
            M("64:79", VA, VB) "v_cvt_f32_i32 v8, v96\n v_cvt_f32_i32 v9, v97\n v_cvt_f32_i32 v10, v98\n v_cvt_f32_i32 v11, v99\n"
            M("80:95", VA, VB) "v_cvt_f32_i32 v12, v112\n v_cvt_f32_i32 v13, v113\n v_cvt_f32_i32 v14, v114\n v_cvt_f32_i32 v15, v115\n"
            M("96:111", VA, VB) "v_cvt_f32_i32 v8, v64\n v_cvt_f32_i32 v9, v65\n v_cvt_f32_i32 v10, v66\n v_cvt_f32_i32 v11, v67\n"
            M("112:127", VA, VB) "v_cvt_f32_i32 v12, v80\n v_cvt_f32_i32 v13, v81\n v_cvt_f32_i32 v14, v82\n v_cvt_f32_i32 v15, v83\n" ::: CLOB);
        if (V == 2) asm volatile(
            M("64:79", VA, VB) "v_lshl_add_u32 v8, v96, 8, v112\n v_lshl_add_u32 v9, v97, 8, v113\n v_lshl_add_u32 v10, v98, 8, v114\n v_lshl_add_u32 v11, v99, 8, v115\n"
            M("80:95", VA, VB) "v_lshl_add_u32 v12, v112, 8, v64\n v_lshl_add_u32 v13, v113, 8, v65\n v_lshl_add_u32 v14, v114, 8, v66\n v_lshl_add_u32 v15, v115, 8, v67\n"
            M("96:111", VA, VB) "v_lshl_add_u32 v8, v64, 8, v80\n v_lshl_add_u32 v9, v65, 8, v81\n v_lshl_add_u32 v10, v66, 8, v82\n v_lshl_add_u32 v11, v67, 8, v83\n"
            M("112:127", VA, VB) "v_lshl_add_u32 v12, v80, 8, v96\n v_lshl_add_u32 v13, v81, 8, v97\n v_lshl_add_u32 v14, v82, 8, v98\n v_lshl_add_u32 v15, v83, 8, v99\n" ::: CLOB);
        if (V == 3) asm volatile(
            M("64:79", AA, AB) "v_cvt_f32_i32 v8, v96\n v_cvt_f32_i32 v9, v97\n v_cvt_f32_i32 v10, v98\n v_cvt_f32_i32 v11, v99\n"
            M("80:95", AA, AB) "v_cvt_f32_i32 v12, v112\n v_cvt_f32_i32 v13, v113\n v_cvt_f32_i32 v14, v114\n v_cvt_f32_i32 v15, v115\n"
            M("96:111", AA, AB) "v_cvt_f32_i32 v8, v64\n v_cvt_f32_i32 v9, v65\n v_cvt_f32_i32 v10, v66\n v_cvt_f32_i32 v11, v67\n"
            M("112:127", AA, AB) "v_cvt_f32_i32 v12, v80\n v_cvt_f32_i32 v13, v81\n v_cvt_f32_i32 v14, v82\n v_cvt_f32_i32 v15, v83\n" ::: CLOB);
        if (V == 4) asm volatile(
            M("64:79", VA, "v[16:19]") "v_cvt_f32_i32 v8, v96\n v_cvt_f32_i32 v9, v97\n v_cvt_f32_i32 v10, v98\n v_cvt_f32_i32 v11, v99\n"
            M("80:95", VA, "v[16:19]") "v_cvt_f32_i32 v12, v112\n v_cvt_f32_i32 v13, v113\n v_cvt_f32_i32 v14, v114\n v_cvt_f32_i32 v15, v115\n"
            M("96:111", VA, "v[8:11]") "v_cvt_f32_i32 v16, v64\n v_cvt_f32_i32 v17, v65\n v_cvt_f32_i32 v18, v66\n v_cvt_f32_i32 v19, v67\n"
            M("112:127", VA, "v[12:15]") "v_cvt_f32_i32 v16, v80\n v_cvt_f32_i32 v17, v81\n v_cvt_f32_i32 v18, v82\n v_cvt_f32_i32 v19, v83\n" ::: CLOB);
        if (V == 5) asm volatile(M("64:79", VA, VB) M("80:95", VA, VB) M("96:111", VA, VB) M("112:127", VA, VB) ::: CLOB);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
void run(long long *c, const char *what) {
    const int iters = 4000;
    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(probe<V>, dim3(256), dim3(256), 0, 0, c, iters);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-100s %6.1f cycles per MFMA\n", what, (double)h / (4.0 * iters));
    fflush(stdout);
}

int main() {
    long long *c;
    hipMalloc(&c, 64);
    run<5>(c, "no fillers");
    run<0>(c, "4 x v_fma_f32 on private registers");
    run<1>(c, "4 x v_cvt_f32_i32 reading accumulator registers (tuple written two MFMAs ago)");
    run<2>(c, "4 x v_lshl_add_u32 reading two accumulator tuples");
    run<3>(c, "4 x v_cvt_f32_i32 reading accumulators, MFMA A / B operands in AccVGPRs");
    run<4>(c, "4 x v_cvt_f32_i32 reading accumulators and writing later MFMAs' B operands");
    return 0;
}
