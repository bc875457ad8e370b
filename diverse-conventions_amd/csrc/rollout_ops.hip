// rollout_ops.hip -- runner-side ops next to the env step, gfx950.
//
// oc_gae_returns: one reverse scan over the whole trajectory batch instead of the reference's Python loop
// of ~10 small torch kernels per time step (train/MAPPO/utils/shared_buffer.py:248-304, the GAE
// branches of SharedReplayBuffer.compute_returns) plus the advantage subtraction of
// train/MAPPO/r_mappo.py:174-177 / train/XD/xd.py:233-239.  f32, memory bound: 12 B read + 4..8 B written
// per (step, env, agent).  Every arithmetic step is an explicitly rounded f32 op in the reference's
// evaluation order (no FMA contraction), so results are bit-identical to the torch expressions.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "overcooked_ops.h"
#include "oc_internal.h"

// Build with -ffp-contract=off (native.build does): __fadd_rn & co. are plain operators in HIP, and an FMA
// rounds once where the reference's separate torch ops round twice.


namespace {


struct GaeArgs {
    const float *rewards;       // [T][M]
    const float *values;        // [T+1][M]   (next_value already stored in row T)
    const float *masks;         // [T+1][M]
    const float *bad_masks;     // [T+1][M] or NULL
    float *returns;             // [T+1][M]   rows 0..T-1 written
    float *advantages;          // [T][M] or NULL
    long long T, M;
    float gamma, gamma_lambda;  // f32(gamma), f32(gamma * gae_lambda) -- the product is taken in double like Python does
    int denormalize;            // values are ValueNorm/PopArt-normalised: v * std + mean
    float vn_mean, vn_std;
    int proper_time_limits;
};

__device__ __forceinline__ float denorm(const GaeArgs &a, float v) {
    return a.denormalize ? __fadd_rn(__fmul_rn(v, a.vn_std), a.vn_mean) : v;      // valuenorm.py:81-83
}

// One thread per (env, agent) column, coalesced across columns at every step.  The scan is serial in t, and
// loads and stores retire through one in-order counter (vmcnt), so a load waited for behind younger stores waits
// for those stores too.  The loop therefore runs in rounds of kRound steps with the NEXT round's inputs issued
// before the current round's results are stored: the wait for them is a counted vmcnt(#stores) that the
// compiler can emit because each round is straight-line code.
constexpr int kRound = 4;

struct GaeIn { float r, m, v, b; };

__device__ __forceinline__ GaeIn gae_load(const GaeArgs &a, long long t, long long m) {
    GaeIn x;
    x.r = a.rewards[t * a.M + m];
    x.m = a.masks[(t + 1) * a.M + m];
    x.v = a.values[t * a.M + m];
    x.b = a.proper_time_limits ? a.bad_masks[(t + 1) * a.M + m] : 1.0f;
    return x;
}

__device__ __forceinline__ void gae_step(const GaeArgs &a, const GaeIn &x, long long t, long long m, float &gae, float &v_next) {
    const float v = denorm(a, x.v);
    // delta = r + gamma * v[t+1] * mask[t+1] - v[t]                          (shared_buffer.py:262-264, 291-293)
    const float delta = __fsub_rn(__fadd_rn(x.r, __fmul_rn(__fmul_rn(a.gamma, v_next), x.m)), v);
    if (a.proper_time_limits) {
        // gae = delta + gamma*lambda * gae * mask ; gae *= bad_mask            (shared_buffer.py:265-266)
        // (the un-normalised branch orders it gamma*lambda * mask * gae, :272 -- both orders are kept)
        const float prod = a.denormalize ? __fmul_rn(__fmul_rn(a.gamma_lambda, gae), x.m)
                                         : __fmul_rn(__fmul_rn(a.gamma_lambda, x.m), gae);
        gae = __fmul_rn(__fadd_rn(delta, prod), x.b);
    } else {
        // gae = delta + gamma*lambda * mask * gae                              (shared_buffer.py:294, 299)
        gae = __fadd_rn(delta, __fmul_rn(__fmul_rn(a.gamma_lambda, x.m), gae));
    }
    const float ret = __fadd_rn(gae, v);
    a.returns[t * a.M + m] = ret;
    if (a.advantages) a.advantages[t * a.M + m] = __fsub_rn(ret, v);             // r_mappo.py:175-177
    v_next = v;
}

__global__ __launch_bounds__(256) void oc_gae_kernel(const GaeArgs a) {
    const long long m = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (m >= a.M) return;
    float gae = 0.0f;
    float v_next = denorm(a, a.values[a.T * a.M + m]);
    long long t = a.T - 1;
    if (t >= kRound - 1) {
        GaeIn cur[kRound], nxt[kRound];
#pragma unroll
        for (int i = 0; i < kRound; ++i) cur[i] = gae_load(a, t - i, m);
        for (; t >= 2 * kRound - 1; t -= kRound) {
#pragma unroll
            for (int i = 0; i < kRound; ++i) nxt[i] = gae_load(a, t - kRound - i, m);      // next round, in flight
#pragma unroll
            for (int i = 0; i < kRound; ++i) gae_step(a, cur[i], t - i, m, gae, v_next);
#pragma unroll
            for (int i = 0; i < kRound; ++i) cur[i] = nxt[i];
        }
#pragma unroll
        for (int i = 0; i < kRound; ++i) gae_step(a, cur[i], t - i, m, gae, v_next);        // last full round
        t -= kRound;
    }
    for (; t >= 0; --t) gae_step(a, gae_load(a, t, m), t, m, gae, v_next);                  // remainder (< kRound steps)
}

}  // namespace

extern "C" int oc_gae_returns(const float *d_rewards, const float *d_value_preds, const float *d_masks,
                              const float *d_bad_masks, float *d_returns, float *d_advantages, int64_t T, int64_t M,
                              double gamma, double gae_lambda, int denormalize, float vn_mean, float vn_std,
                              int proper_time_limits, void *stream) {
    if (!d_rewards || !d_value_preds || !d_masks || !d_returns || T < 0 || M < 0) return OC_ERR_INVALID_ARG;
    if (proper_time_limits && !d_bad_masks) return OC_ERR_INVALID_ARG;
    if (T == 0 || M == 0) return OC_OK;
    GaeArgs a;
    a.rewards = d_rewards; a.values = d_value_preds; a.masks = d_masks; a.bad_masks = d_bad_masks;
    a.returns = d_returns; a.advantages = d_advantages;
    a.T = T; a.M = M;
    a.gamma = static_cast<float>(gamma);
    a.gamma_lambda = static_cast<float>(gamma * gae_lambda);
    a.denormalize = denormalize; a.vn_mean = vn_mean; a.vn_std = vn_std;
    a.proper_time_limits = proper_time_limits;
    const OcDeviceScope on_device(oc_device_of_pointer(d_returns));
    const unsigned grid = static_cast<unsigned>((M + 255) / 256);
    hipLaunchKernelGGL(oc_gae_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}

// ======================================================================================================
// Advantage normalisation (train/XD/xd.py:240-244 = train/MAPPO/r_mappo.py:177-181).
//   copy[active == 0] = nan ; mean = nanmean(copy) ; std = copy[copy != nan].std() ; adv = (adv - mean) / (std + 1e-5)
// `copy != nan` is true for every element, NaN included, so the reference's std is over ALL entries and is NaN
// as soon as one entry is inactive; std_over_active = 0 reproduces that, 1 takes the std of the active entries.
// Two launches: per-block partial sums in f64 (fixed grid, fixed order -> run-to-run reproducible), then one
// block that folds the partials, writes {mean, std} and -- in the same launch sequence -- the element-wise pass.
// HBM bound: 8 B read (4 without a mask) per element for the statistics, 4 B read + 4 B written for the pass.
// ======================================================================================================
namespace {

constexpr int kStatBlocks = 1024;

struct AdvPartial {
    double sum, sumsq;
    long long n_active, n_inactive;
};

__device__ __forceinline__ void adv_accumulate(float x, float act, AdvPartial &p) {
    if (act != 0.0f) {
        p.sum += static_cast<double>(x);
        p.sumsq += static_cast<double>(x) * static_cast<double>(x);
        p.n_active += 1;
    } else {
        p.n_inactive += 1;
    }
}

__device__ __forceinline__ AdvPartial adv_block_reduce(AdvPartial p) {
    __shared__ AdvPartial part[4];
    for (int off = 32; off > 0; off >>= 1) {
        p.sum += __shfl_down(p.sum, off);
        p.sumsq += __shfl_down(p.sumsq, off);
        p.n_active += __shfl_down(p.n_active, off);
        p.n_inactive += __shfl_down(p.n_inactive, off);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) part[wave] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            p.sum += part[w].sum;
            p.sumsq += part[w].sumsq;
            p.n_active += part[w].n_active;
            p.n_inactive += part[w].n_inactive;
        }
    }
    return p;      // valid in thread 0
}

__global__ __launch_bounds__(256) void oc_adv_partial_kernel(const float *adv, const float *active, long long n, AdvPartial *partials) {
    AdvPartial p = {0.0, 0.0, 0, 0};
    const long long n4 = n >> 2;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const float4 *adv4 = reinterpret_cast<const float4 *>(adv);
    const float4 *act4 = reinterpret_cast<const float4 *>(active);
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = adv4[i];
        const float4 m = active ? act4[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        adv_accumulate(x.x, m.x, p);
        adv_accumulate(x.y, m.y, p);
        adv_accumulate(x.z, m.z, p);
        adv_accumulate(x.w, m.w, p);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                      // the last n % 4 elements
        const long long i = (n4 << 2) + threadIdx.x;
        adv_accumulate(adv[i], active ? active[i] : 1.0f, p);
    }
    p = adv_block_reduce(p);
    if (threadIdx.x == 0) partials[blockIdx.x] = p;
}

__global__ __launch_bounds__(256) void oc_adv_finalize_kernel(const AdvPartial *partials, int n_partials, int std_over_active, float *stats) {
    AdvPartial p = {0.0, 0.0, 0, 0};
    for (int i = threadIdx.x; i < n_partials; i += 256) {
        p.sum += partials[i].sum;
        p.sumsq += partials[i].sumsq;
        p.n_active += partials[i].n_active;
        p.n_inactive += partials[i].n_inactive;
    }
    p = adv_block_reduce(p);
    if (threadIdx.x == 0) {
        const double n = static_cast<double>(p.n_active);
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        const double mean = p.n_active > 0 ? p.sum / n : nan;                                   // torch.nanmean
        double var = nan;
        if (p.n_active > 1 && (std_over_active || p.n_inactive == 0)) {
            var = (p.sumsq - p.sum * mean) / (n - 1.0);                                       // unbiased, torch .std()
            if (var < 0.0) var = 0.0;
        }
        stats[0] = static_cast<float>(mean);
        stats[1] = static_cast<float>(sqrt(var));
    }
}

__global__ __launch_bounds__(256) void oc_adv_apply_kernel(const float *adv, const float *stats, long long n, float *out) {
    const float mean = stats[0];
    const float denom = __fadd_rn(stats[1], 1e-5f);                                            // std + 1e-5 in f32, as torch does
    const long long n4 = n >> 2;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const float4 *in4 = reinterpret_cast<const float4 *>(adv);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 x = in4[i];
        x.x = __fdiv_rn(__fsub_rn(x.x, mean), denom);
        x.y = __fdiv_rn(__fsub_rn(x.y, mean), denom);
        x.z = __fdiv_rn(__fsub_rn(x.z, mean), denom);
        x.w = __fdiv_rn(__fsub_rn(x.w, mean), denom);
        out4[i] = x;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        out[i] = __fdiv_rn(__fsub_rn(adv[i], mean), denom);
    }
}

}  // namespace

extern "C" int64_t oc_advantage_workspace_bytes(void) { return static_cast<int64_t>(kStatBlocks) * sizeof(AdvPartial); }

extern "C" int oc_advantage_normalize(const float *d_adv, const float *d_active_masks, int64_t n, int std_over_active, float *d_out,
                                      float *d_stats, void *d_workspace, void *stream) {
    if (!d_adv || !d_stats || !d_workspace || n <= 0) return OC_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(d_adv) & 15) || (reinterpret_cast<uintptr_t>(d_active_masks) & 15) ||
        (reinterpret_cast<uintptr_t>(d_out) & 15) || (reinterpret_cast<uintptr_t>(d_workspace) & 7))
        return OC_ERR_INVALID_ARG;
    const OcDeviceScope on_device(oc_device_of_pointer(d_adv));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n4 = n >> 2;
    int blocks = static_cast<int>(std::min<long long>(kStatBlocks, std::max<long long>(1, (n4 + 255) / 256)));
    AdvPartial *partials = static_cast<AdvPartial *>(d_workspace);
    hipLaunchKernelGGL(oc_adv_partial_kernel, dim3(blocks), dim3(256), 0, s, d_adv, d_active_masks, static_cast<long long>(n), partials);
    hipLaunchKernelGGL(oc_adv_finalize_kernel, dim3(1), dim3(256), 0, s, partials, blocks, std_over_active, d_stats);
    if (d_out) {
        const int apply_blocks = static_cast<int>(std::min<long long>(4096, std::max<long long>(1, (n4 + 255) / 256)));
        hipLaunchKernelGGL(oc_adv_apply_kernel, dim3(apply_blocks), dim3(256), 0, s, d_adv, d_stats, static_cast<long long>(n), d_out);
    }
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}

// ======================================================================================================
// Row gather: dst row i <- src row index[i].  One kernel behind the minibatch generators (obs rows picked by a
// permutation, train/MAPPO/utils/shared_buffer.py:331-345), the per-agent generators (:393-407) and the
// diagonal / partial buffer inserts (:167-179, :200-212) once their index arithmetic is done on the device.
// Rows are byte rows (int8 observations: S*C bytes, a multiple of 4); the destination is either the same bytes or
// f32 (the reference's buffer dtype) with the cast fused.  HBM bound: row_bytes read + row_bytes (x4 for f32)
// written + 8 B of index per row.
//
// A wavefront takes 64 rows at a time: lane l holds index[base + l]; the 64 * units_per_row transfer units of
// the batch are dealt to lanes flat, so every store instruction covers 64 consecutive 16-byte chunks of the
// destination whatever the row length, and the source row of a unit comes from a cross-lane read of the index.
// ======================================================================================================
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GatherArgs {
    const uint8_t *src;
    uint8_t *dst;
    const long long *index;     // [n_rows] source row per destination row; NULL = identity
    long long n_rows, src_rows;
    int row_bytes;              // source row length
    int units;                  // transfer units per row (16-byte chunks, or dwords)
    float inv_units;
};

enum { GATHER_COPY16 = 0, GATHER_COPY4 = 1, GATHER_I8_TO_F32 = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void oc_gather_rows_kernel(const GatherArgs a) {
    const int lane = threadIdx.x & 63;
    const long long wave = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const long long n_waves = static_cast<long long>(gridDim.x) * 4;
    const long long n_batches = (a.n_rows + 63) >> 6;
    constexpr int SRC_UNIT = MODE == GATHER_COPY16 ? 16 : 4;
    constexpr int DST_UNIT = MODE == GATHER_COPY4 ? 4 : 16;
    for (long long b = wave; b < n_batches; b += n_waves) {
        const long long base = b << 6;
        const int rows = static_cast<int>(a.n_rows - base < 64 ? a.n_rows - base : 64);
        long long r = base + (lane < rows ? lane : 0);
        if (a.index) r = a.index[r];
        r = r < 0 ? 0 : (r >= a.src_rows ? a.src_rows - 1 : r);              // never fault on a bad index
        const int lo = static_cast<int>(r), hi = static_cast<int>(r >> 32);
        const int total = rows * a.units;
        uint8_t *out = a.dst + base * static_cast<long long>(a.units) * DST_UNIT;
        // Four units per lane in flight (round 5): all four loads -- unconditional, the unit index clamped into the batch -- are issued
        // before the first store.  As one load / one store per iteration (rounds 1-4) a wavefront had a single kilobyte in flight and
        // waited out a memory round trip per unit: 0.52 of the HBM roofline on 400-byte rows (profiles/r01_bench_minibatch.log).
        constexpr int kIlp = 4;
        for (int c0 = 0; c0 < total; c0 += 64 * kIlp) {                      // wave-uniform trip count: the cross-lane reads
            u32x4 v16[kIlp];                                                 // below need every lane alive
            uint32_t v4[kIlp];
            int cs[kIlp];
#pragma unroll
            for (int j = 0; j < kIlp; ++j) {
                const int c = c0 + j * 64 + lane;
                cs[j] = c;
                const int cc = min(c, total - 1);
                // row of unit cc within the batch: cc < 64 * 1280, so the float quotient is exact (error < 0.5 / units)
                const int row = min(static_cast<int>((static_cast<float>(cc) + 0.5f) * a.inv_units), 63);
                const int k = cc - row * a.units;
                const long long sr = (static_cast<long long>(__shfl(hi, row)) << 32) | static_cast<uint32_t>(__shfl(lo, row));
                const uint8_t *in = a.src + sr * a.row_bytes + static_cast<long long>(k) * SRC_UNIT;
                if (MODE == GATHER_COPY16) v16[j] = *reinterpret_cast<const u32x4 *>(in);
                else v4[j] = *reinterpret_cast<const uint32_t *>(in);
            }
#pragma unroll
            for (int j = 0; j < kIlp; ++j) {
                const int c = cs[j];
                if (c >= total) continue;
                if (MODE == GATHER_COPY16) {
                    *reinterpret_cast<u32x4 *>(out + static_cast<long long>(c) * 16) = v16[j];
                } else if (MODE == GATHER_COPY4) {
                    *reinterpret_cast<uint32_t *>(out + static_cast<long long>(c) * 4) = v4[j];
                } else {
                    const uint32_t w = v4[j];
                    f32x4 f;
                    f.x = static_cast<float>(static_cast<int8_t>(w));
                    f.y = static_cast<float>(static_cast<int8_t>(w >> 8));
                    f.z = static_cast<float>(static_cast<int8_t>(w >> 16));
                    f.w = static_cast<float>(static_cast<int8_t>(w >> 24));
                    *reinterpret_cast<f32x4 *>(out + static_cast<long long>(c) * 16) = f;
                }
            }
        }
    }
}

// Scalar columns of a minibatch (value_preds, returns, masks, active_masks, actions, action_log_probs, advantages,
// rnn_states ...): up to 8 f32 columns of `words` 4-byte words per row gathered in one launch with one index read.
// Source row = index[i] * row_mul + row_add: (1, 0) for feed_forward_generator, (A, agent) for partial_ff_generator.
struct ColumnArgs {
    const uint32_t *src[8];
    uint32_t *dst[8];
    int words[8];
    long long src_rows[8];
    const float *norm_stats[8];   // {mean, std}: (x - mean) / (std + 1e-5) applied on the fly, or NULL
    int n_cols;
    const long long *index;
    long long n_rows, row_mul, row_add;
};

__global__ __launch_bounds__(256) void oc_gather_columns_kernel(const ColumnArgs a) {
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n_rows; i += stride) {
        const long long r0 = a.index[i] * a.row_mul + a.row_add;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= a.n_cols) break;
            const long long r = r0 < 0 ? 0 : (r0 >= a.src_rows[c] ? a.src_rows[c] - 1 : r0);
            const int w = a.words[c];
            if (a.norm_stats[c]) {
                const float mean = a.norm_stats[c][0], denom = __fadd_rn(a.norm_stats[c][1], 1e-5f);
                for (int k = 0; k < w; ++k)
                    a.dst[c][i * w + k] = __float_as_uint(__fdiv_rn(__fsub_rn(__uint_as_float(a.src[c][r * w + k]), mean), denom));
            } else {
                for (int k = 0; k < w; ++k) a.dst[c][i * w + k] = a.src[c][r * w + k];
            }
        }
    }
}

}  // namespace

extern "C" int oc_gather_rows(const void *d_src, int64_t src_rows, int32_t row_bytes, const int64_t *d_index, int64_t n_rows, void *d_dst,
                              int dst_dtype, void *stream) {
    if (!d_src || !d_dst || src_rows <= 0 || n_rows < 0 || row_bytes <= 0 || (row_bytes & 3) || row_bytes > 5120)
        return OC_ERR_INVALID_ARG;
    if (dst_dtype != 0 && dst_dtype != 1) return OC_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(d_src) & 15) || (reinterpret_cast<uintptr_t>(d_dst) & 15) || (reinterpret_cast<uintptr_t>(d_index) & 7))
        return OC_ERR_INVALID_ARG;
    if (n_rows == 0) return OC_OK;
    GatherArgs a;
    a.src = static_cast<const uint8_t *>(d_src);
    a.dst = static_cast<uint8_t *>(d_dst);
    a.index = reinterpret_cast<const long long *>(d_index);
    a.n_rows = n_rows;
    a.src_rows = src_rows;
    a.row_bytes = row_bytes;
    const int mode = dst_dtype == 1 ? GATHER_I8_TO_F32 : ((row_bytes & 15) == 0 ? GATHER_COPY16 : GATHER_COPY4);
    a.units = mode == GATHER_COPY16 ? row_bytes / 16 : row_bytes / 4;
    a.inv_units = 1.0f / static_cast<float>(a.units);
    const OcDeviceScope on_device(oc_device_of_pointer(d_dst));
    const long long n_batches = (n_rows + 63) >> 6;
    const int grid = static_cast<int>(std::min<long long>(256 * 8, (n_batches + 3) / 4));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == GATHER_COPY16) hipLaunchKernelGGL(oc_gather_rows_kernel<GATHER_COPY16>, dim3(grid), dim3(256), 0, s, a);
    else if (mode == GATHER_COPY4) hipLaunchKernelGGL(oc_gather_rows_kernel<GATHER_COPY4>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(oc_gather_rows_kernel<GATHER_I8_TO_F32>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}

extern "C" int oc_gather_columns(int32_t n_cols, const void *const *d_src, const int64_t *src_rows, const int32_t *words, void *const *d_dst,
                                 const float *const *d_norm_stats, const int64_t *d_index, int64_t n_rows, int64_t row_mul, int64_t row_add,
                                 void *stream) {
    if (n_cols < 1 || n_cols > 8 || !d_src || !src_rows || !words || !d_dst || !d_index || n_rows < 0 || row_mul < 1 || row_add < 0)
        return OC_ERR_INVALID_ARG;
    if (n_rows == 0) return OC_OK;
    ColumnArgs a = {};
    for (int c = 0; c < n_cols; ++c) {
        if (!d_src[c] || !d_dst[c] || words[c] < 1 || src_rows[c] < 1) return OC_ERR_INVALID_ARG;
        if ((reinterpret_cast<uintptr_t>(d_src[c]) & 3) || (reinterpret_cast<uintptr_t>(d_dst[c]) & 3)) return OC_ERR_INVALID_ARG;
        a.src[c] = static_cast<const uint32_t *>(d_src[c]);
        a.dst[c] = static_cast<uint32_t *>(d_dst[c]);
        a.words[c] = words[c];
        a.src_rows[c] = src_rows[c];
        a.norm_stats[c] = d_norm_stats ? d_norm_stats[c] : nullptr;
    }
    a.n_cols = n_cols;
    a.index = reinterpret_cast<const long long *>(d_index);
    a.n_rows = n_rows;
    a.row_mul = row_mul;
    a.row_add = row_add;
    const OcDeviceScope on_device(oc_device_of_pointer(d_index));
    const int grid = static_cast<int>(std::min<long long>(256 * 8, (n_rows + 255) / 256));
    hipLaunchKernelGGL(oc_gather_columns_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}

// ======================================================================================================
// Sample records.  A minibatch reads seven one-word columns per sample (actions, value_preds, returns, masks,
// active_masks, action_log_probs, advantages) at a random row: seven 32-byte sectors for 28 useful bytes.
// oc_pack_columns interleaves up to eight columns into one 32-byte record per sample, once per update
// (streaming, coalesced); oc_gather_records then reads ONE sector per sample and writes the columns out
// separately and contiguously, as the reference's generators yield them.
// ======================================================================================================
namespace {

struct PackArgs {
    const uint32_t *src[8];
    int n_cols;
    long long n_rows;
    u32x4 *packed;              // [n_rows][2]
};

__global__ __launch_bounds__(256) void oc_pack_columns_kernel(const PackArgs a) {
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n_rows; i += stride) {
        uint32_t w[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = c < a.n_cols ? a.src[c][i] : 0u;
        u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
        a.packed[2 * i] = lo;
        a.packed[2 * i + 1] = hi;
    }
}

struct RecordArgs {
    const u32x4 *packed;
    uint32_t *dst[8];
    int n_cols, norm_col;       // norm_col < 0: none
    const float *norm_stats;
    const long long *index;
    long long n_rows, src_rows, row_mul, row_add;
};

__global__ __launch_bounds__(256) void oc_gather_records_kernel(const RecordArgs a) {
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    float mean = 0.0f, denom = 1.0f;
    if (a.norm_col >= 0) {
        mean = a.norm_stats[0];
        denom = __fadd_rn(a.norm_stats[1], 1e-5f);
    }
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n_rows; i += stride) {
        long long r = a.index[i] * a.row_mul + a.row_add;
        r = r < 0 ? 0 : (r >= a.src_rows ? a.src_rows - 1 : r);
        const u32x4 lo = a.packed[2 * r], hi = a.packed[2 * r + 1];
        uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= a.n_cols) break;
            uint32_t v = w[c];
            if (c == a.norm_col) v = __float_as_uint(__fdiv_rn(__fsub_rn(__uint_as_float(v), mean), denom));
            a.dst[c][i] = v;
        }
    }
}

}  // namespace

extern "C" int oc_pack_columns(int32_t n_cols, const void *const *d_src, int64_t n_rows, void *d_packed, void *stream) {
    if (n_cols < 1 || n_cols > 8 || !d_src || !d_packed || n_rows < 0 || (reinterpret_cast<uintptr_t>(d_packed) & 15)) return OC_ERR_INVALID_ARG;
    if (n_rows == 0) return OC_OK;
    PackArgs a = {};
    for (int c = 0; c < n_cols; ++c) {
        if (!d_src[c] || (reinterpret_cast<uintptr_t>(d_src[c]) & 3)) return OC_ERR_INVALID_ARG;
        a.src[c] = static_cast<const uint32_t *>(d_src[c]);
    }
    a.n_cols = n_cols;
    a.n_rows = n_rows;
    a.packed = static_cast<u32x4 *>(d_packed);
    const OcDeviceScope on_device(oc_device_of_pointer(d_packed));
    const int grid = static_cast<int>(std::min<long long>(256 * 8, (n_rows + 255) / 256));
    hipLaunchKernelGGL(oc_pack_columns_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}

extern "C" int oc_gather_records(const void *d_packed, int64_t src_rows, int32_t n_cols, void *const *d_dst, int32_t norm_col,
                                 const float *d_norm_stats, const int64_t *d_index, int64_t n_rows, int64_t row_mul, int64_t row_add,
                                 void *stream) {
    if (!d_packed || src_rows < 1 || n_cols < 1 || n_cols > 8 || !d_dst || !d_index || n_rows < 0 || row_mul < 1 || row_add < 0 ||
        norm_col >= n_cols || (norm_col >= 0 && !d_norm_stats) || (reinterpret_cast<uintptr_t>(d_packed) & 15))
        return OC_ERR_INVALID_ARG;
    if (n_rows == 0) return OC_OK;
    RecordArgs a = {};
    a.packed = static_cast<const u32x4 *>(d_packed);
    for (int c = 0; c < n_cols; ++c) {
        if (!d_dst[c] || (reinterpret_cast<uintptr_t>(d_dst[c]) & 3)) return OC_ERR_INVALID_ARG;
        a.dst[c] = static_cast<uint32_t *>(d_dst[c]);
    }
    a.n_cols = n_cols;
    a.norm_col = norm_col < 0 ? -1 : norm_col;
    a.norm_stats = d_norm_stats;
    a.index = reinterpret_cast<const long long *>(d_index);
    a.n_rows = n_rows;
    a.src_rows = src_rows;
    a.row_mul = row_mul;
    a.row_add = row_add;
    const OcDeviceScope on_device(oc_device_of_pointer(d_packed));
    const int grid = static_cast<int>(std::min<long long>(256 * 8, (n_rows + 255) / 256));
    hipLaunchKernelGGL(oc_gather_records_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}

// ---- stable two-way partition of row ids (mixed play) -------------------------------------------------
// The reference picks, per environment and step, which of two policies acts and evaluates each policy on the rows it was given
// (`mix_mask == i` indexing, train/partner_agents.py:171-196, train/XD/xd_player.py:287-305).  oc_partition_rows turns a batch
// of such 0/1 choice vectors into index lists the policy kernel's gather tiles read (oc_policy_forward_gather): per list, the ids
// of the entries with select == 0 in ascending order, then the ids of the others, and both counts.  One workgroup per list, two
// passes over it (count, then place) with a wavefront-ballot scan per 1024-entry chunk.
namespace {

__global__ __launch_bounds__(1024) void oc_partition_rows_kernel(const int32_t *select, long long len, long long id_period, int32_t *index,
                                                                 int32_t *count) {
    __shared__ int wave_zeros[16];
    __shared__ int total_zeros;
    const long long list = blockIdx.x;
    const int32_t *sel = select + list * len;
    int32_t *out = index + list * len;
    const int id0 = static_cast<int>((list % id_period) * len);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 1: zeros in the whole list
    int zeros = 0;
    for (long long k = threadIdx.x; k < len; k += 1024) zeros += sel[k] == 0 ? 1 : 0;
    for (int d = 32; d > 0; d >>= 1) zeros += __shfl_down(zeros, d);
    if (lane == 0) wave_zeros[wave] = zeros;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wave_zeros[w];
        total_zeros = t;
        count[2 * list] = t;
        count[2 * list + 1] = static_cast<int>(len) - t;
    }
    __syncthreads();
    const int c0 = total_zeros;
    // pass 2: place.  run0 / run1: entries of either class placed by the chunks before this one
    int run0 = 0;
    for (long long base = 0; base < len; base += 1024) {
        const long long k = base + threadIdx.x;
        const bool in = k < len;
        const bool zero = in && sel[k] == 0;
        const unsigned long long bz = __ballot(zero);
        const int before = __popcll(bz & ((1ull << lane) - 1ull));           // zeros of this wavefront below this lane
        __syncthreads();                                                       // wave_zeros of the previous chunk has been read
        if (lane == 0) wave_zeros[wave] = __popcll(bz);
        __syncthreads();
        int zeros_below = 0, zeros_chunk = 0;
        for (int w = 0; w < 16; ++w) {
            zeros_below += w < wave ? wave_zeros[w] : 0;
            zeros_chunk += wave_zeros[w];
        }
        if (in) {
            const int z_before = run0 + zeros_below + before;                  // zeros among entries [0, k)
            const int pos = zero ? z_before : c0 + (static_cast<int>(k) - z_before);
            out[pos] = id0 + static_cast<int>(k);
        }
        run0 += zeros_chunk;
    }
}

}  // namespace

extern "C" int oc_partition_rows(const int32_t *d_select, int64_t n_lists, int64_t len, int64_t id_period, int32_t *d_index, int32_t *d_count,
                                 void *stream) {
    if (!d_select || !d_index || !d_count || n_lists < 0 || len < 1 || id_period < 1 || len * id_period > INT32_MAX || n_lists > INT32_MAX)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_partition_rows: NULL pointer, or sizes outside 1 .. 2^31 - 1 (len * id_period included)");
    if (n_lists == 0) return OC_OK;
    const OcDeviceScope on_device(oc_device_of_pointer(d_index));
    hipLaunchKernelGGL(oc_partition_rows_kernel, dim3(static_cast<unsigned>(n_lists)), dim3(1024), 0, static_cast<hipStream_t>(stream), d_select,
                       static_cast<long long>(len), static_cast<long long>(id_period), d_index, d_count);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return oc_internal_fail(OC_ERR_HIP, "oc_partition_rows: launch failed: %s", hipGetErrorString(err));
    return OC_OK;
}
