// split_bf16.hip -- is the dot2c form of the exact three-way bf16 split bit-identical to the AND / subtract form?
//
// The policy kernel splits every f32 activation into three bf16 pieces by truncation: x = hi + mid + lo exactly
// (policy_ops.hip, split_pair).  Old form per pair of values: 2 AND + 1 packed subtract per level + 3 byte permutes = 9 VALU.
// New form: the permute that packs the two truncated values comes first, and the remainder of each value is one
// v_dot2c_f32_bf16 (acc += a.lo * b.lo + a.hi * b.hi) against the constant {-1, 0} / {0, -1}: x - trunc(x), exactly
// representable, so whatever rounding the instruction applies cannot show -- 7 VALU.  This probe checks that claim on the
// hardware for 2^26 random bit patterns plus edge cases (zeros, denormals, powers of two, values whose low bits are all ones),
// and times both forms.
//   hipcc --offload-arch=gfx950 -O3 -o split_bf16 split_bf16.hip && ./split_bf16
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_old(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const v2f xv = {x0, x1};
    const v2f hv = {__uint_as_float(__float_as_uint(x0) & 0xffff0000u), __uint_as_float(__float_as_uint(x1) & 0xffff0000u)};
    const v2f rv = xv - hv;
    const v2f mv = {__uint_as_float(__float_as_uint(rv.x) & 0xffff0000u), __uint_as_float(__float_as_uint(rv.y) & 0xffff0000u)};
    const v2f qv = rv - mv;
    hi = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
    mid = __builtin_amdgcn_perm(__float_as_uint(rv.y), __float_as_uint(rv.x), 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(qv.y), __float_as_uint(qv.x), 0x07060302u);
}

__device__ __forceinline__ void split_new(float x0, float x1, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    // {-1, 0} and {0, -1}, kept in SGPRs behind an empty asm: as a plain constant hipcc 7.2 folds 0x0000bf80 into the inline
    // operand "-1.0", which the instruction reads as the f32 pattern 0xbf800000 = {0, -1} -- the other constant
    // (tools/probes/split_bf16.hip caught it: every first element's remainder came out as x0 - trunc(x1))
    uint32_t k_lo = 0x0000bf80u, k_hi = 0xbf800000u;
    asm volatile("" : "+s"(k_lo), "+s"(k_hi));
    const v2bf m_lo = __builtin_bit_cast(v2bf, k_lo), m_hi = __builtin_bit_cast(v2bf, k_hi);
    hi = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
    const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, hi), m_lo, x0, false);
    const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, hi), m_hi, x1, false);
    mid = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
    const float q0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, mid), m_lo, r0, false);
    const float q1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, mid), m_hi, r1, false);
    lo = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

__global__ void check(const uint32_t *bits, int64_t n_pairs, unsigned long long *bad, uint32_t *first_bad) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        const float x0 = __uint_as_float(bits[2 * i]), x1 = __uint_as_float(bits[2 * i + 1]);
        uint32_t a, b, c, d, e, f;
        split_old(x0, x1, a, b, c);
        split_new(x0, x1, d, e, f);
        if (a != d || b != e || c != f) {
            if (atomicAdd(bad, 1ull) == 0) { first_bad[0] = bits[2 * i]; first_bad[1] = bits[2 * i + 1]; first_bad[2] = b; first_bad[3] = e; first_bad[4] = c; first_bad[5] = f; }
        }
    }
}

template <bool NEW>
__global__ void rate(uint32_t *out, int iters) {
    float x0 = 1.0f + threadIdx.x * 1e-3f, x1 = 2.0f + threadIdx.x * 1e-3f;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t a, b, c;
        if (NEW) split_new(x0, x1, a, b, c); else split_old(x0, x1, a, b, c);
        acc ^= a ^ b ^ c;
        x0 = __uint_as_float((__float_as_uint(x0) + (acc & 0xff)) | 0x3f000000u & 0x7f7fffffu);
        x1 = __uint_as_float((__float_as_uint(x1) + (acc & 0xfe)) | 0x3f000000u & 0x7f7fffffu);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int64_t n_pairs = 1ll << 25;
    std::vector<uint32_t> h(2 * n_pairs);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (auto &v : h) {
        uint32_t u = rnd();
        if (((u >> 23) & 0xff) == 0xff) u &= 0xbf7fffffu;                 // finite values only: activations are finite
        v = u;
    }
    const uint32_t edge[] = {0u, 0x80000000u, 1u, 0x007fffffu, 0x00800000u, 0x3f800000u, 0x3f80ffffu, 0x3fffffffu, 0x7f7fffffu, 0xff7fffffu,
                             0x3f7fffffu, 0x00ffffffu, 0x0080ffffu, 0x34000000u, 0x33ffffffu, 0x4b7fffffu};
    for (size_t i = 0; i < sizeof(edge) / 4; ++i)
        for (size_t j = 0; j < sizeof(edge) / 4; ++j) { h[2 * (i * 16 + j)] = edge[i]; h[2 * (i * 16 + j) + 1] = edge[j]; }
    // the activations the kernel really sees: small non-negative numbers, many exact zeros
    for (int64_t i = 1 << 20; i < (1 << 22); ++i) { float f = (rnd() & 3) ? (float)(rnd() & 0xffffff) * (1.0f / 4096.0f) : 0.f; memcpy(&h[i], &f, 4); }
    uint32_t *d; unsigned long long *bad; uint32_t *fb;
    hipMalloc(&d, h.size() * 4); hipMalloc(&bad, 8); hipMalloc(&fb, 32);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 8); hipMemset(fb, 0, 32);
    check<<<2048, 256>>>(d, n_pairs, bad, fb);
    unsigned long long nb = 0; uint32_t f[8] = {};
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 32, hipMemcpyDeviceToHost);
    printf("pairs checked %lld, mismatching %llu", (long long)n_pairs, nb);
    if (nb) printf("  first: x0 %08x x1 %08x mid old %08x new %08x lo old %08x new %08x", f[0], f[1], f[2], f[3], f[4], f[5]);
    printf("\n");
    uint32_t *o; hipMalloc(&o, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (v) rate<true><<<1024, 256>>>(o, 20000); else rate<false><<<1024, 256>>>(o, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s form: %.3f ms for 1024 x 4 waves x 20000 splits\n", v ? "dot2c" : "and/sub", ms);
    }
    return nb ? 1 : 0;
}
