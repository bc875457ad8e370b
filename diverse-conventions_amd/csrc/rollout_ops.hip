// rollout_ops.hip -- runner-side ops next to the env step, gfx950.
//
// oc_gae_returns: one reverse scan over the whole trajectory batch instead of the reference's Python loop
// of ~10 small torch kernels per time step (train/MAPPO/utils/shared_buffer.py:248-304, the GAE
// branches of SharedReplayBuffer.compute_returns) plus the advantage subtraction of
// train/MAPPO/r_mappo.py:174-177 / train/XD/xd.py:233-239.  f32, memory bound: 12 B read + 4..8 B written
// per (step, env, agent).  Every arithmetic step is an explicitly rounded f32 op in the reference's
// evaluation order (no FMA contraction), so results are bit-identical to the torch expressions.

#include <hip/hip_runtime.h>

#include "overcooked_hip.h"

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
    const unsigned grid = static_cast<unsigned>((M + 255) / 256);
    hipLaunchKernelGGL(oc_gae_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? OC_OK : OC_ERR_HIP;
}
