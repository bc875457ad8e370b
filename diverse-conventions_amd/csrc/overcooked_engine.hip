// overcooked_engine.hip -- MI355X (gfx950) Overcooked rollout engine: kernels + C ABI.
//
// What it replaces: the reference's Madrona simulation of the Overcooked MDP
// (/root/reference/src/overcooked2_env/sim.cpp: 13 task-graph nodes per step, :649-688) and its
// Manager/bindings (mgr.cpp, bindings.cpp).  Semantics follow the reference's *Python* twin
// (envs/overcooked2_reimplement.py, DummyMDP) which SURVEY.md 8(a) makes normative; reference
// line numbers below are of that file unless another file is named.
//
// Execution model (see DESIGN.md for the roofline argument):
//   * one 64-lane wavefront (= one workgroup) owns KPW consecutive kitchens;
//   * their state records are flat-copied HBM -> LDS with 16-byte loads (odd LDS stride per
//     kitchen so lane-per-kitchen accesses are bank-conflict free);
//   * phase T (lane = kitchen): the serial joint transition -- interacts in player order, moves,
//     collision rule, timestep/horizon, auto-reset -- entirely on LDS/registers;
//   * phase E (lane = tile): pot timers, reset clearing, and the 20-channel encoding of every tile
//     for both viewers, 5 dwords per (tile, viewer) staged in LDS in *output* order;
//   * flush: the staged planes leave as 16-byte-per-lane, 1 KiB-per-wave coalesced stores; the
//     state block goes back the same way.  No atomics, no inter-workgroup traffic.
//
// This file is the product path.  It never calls into oracle/ and has no CPU fallback.

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "overcooked_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kChannels = 20;            // 5P + 10, P = 2   (ref:165-171)
constexpr int kTileDwords = kChannels / 4;

// object names (ref:4-9), terrain codes (ref:12-19), actions (ref:35-43)
enum : uint32_t { O_NONE = 0, O_TOMATO = 1, O_ONION = 2, O_DISH = 3, O_SOUP = 4 };
enum : uint32_t { T_AIR = 0, T_POT = 1, T_COUNTER = 2, T_ONION_SRC = 3, T_DISH_SRC = 4, T_SERVING = 5, T_TOMATO_SRC = 6 };
enum : int { A_NORTH = 0, A_SOUTH = 1, A_EAST = 2, A_WEST = 3, A_STAY = 4, A_INTERACT = 5 };

// packed object: name | onions<<8 | tomatoes<<16 | (uint8)tick<<24 ; fresh objects carry tick = -1 (ref:46-51)
constexpr uint32_t kFreshTick = 0xFF000000u;
constexpr uint32_t kOnionObj = kFreshTick | O_ONION;
constexpr uint32_t kTomatoObj = kFreshTick | O_TOMATO;
constexpr uint32_t kDishObj = kFreshTick | O_DISH;
constexpr uint32_t kEmptySoup = kFreshTick | O_SOUP;

constexpr int kMaxPots = 32;

// Per-layout lookup tables, resident in HBM (L2-hot), copied to LDS by every wave.
struct DevTables {
    uint32_t terrain[256];      // by state tile index t = y*W + x
    uint32_t out_tab[256];      // by output tile index o = x*H + y :  t | terrain<<8
    int32_t recipe_values[16];  // ref:283-285
    int32_t recipe_times[16];   // ref:156-157
    uint32_t pots[kMaxPots];    // tile indices with terrain POT
    uint32_t counters[256];     // tile indices with terrain COUNTER
};

struct StepArgs {
    uint4 *state;               // [N][recq] uint4
    const DevTables *tabs;
    const void *actions;        // [P][N] i32 or i64
    int8_t *obs;                // viewer v plane at obs + v*obs_player_stride ; [N][W][H][C]
    int32_t *reward;            // [P][N]
    int32_t *done;              // [N]
    int64_t n_kitchens;
    int64_t obs_player_stride;  // bytes
    int32_t W, S;
    int32_t recq;               // uint4 per state record
    int32_t lstride;            // LDS dwords per kitchen = 4*recq + 1 (odd -> conflict free)
    int32_t kpw;                // kitchens per wave
    int32_t gpasses;            // 64-tile passes staged per flush
    int32_t horizon, placement_rew, dish_rew, soup_rew;
    int32_t n_pots, n_counters;
    int32_t uniform_time;       // recipe time if all 16 are equal, else -1
    uint32_t start_posor;       // p0.pos | NORTH<<8 | p1.pos<<16 | NORTH<<24   (ref:387-391)
};

struct LdsView {
    uint32_t *terrain, *out_tab, *pots, *counters;
    int32_t *rvalues, *rtimes;
    uint32_t *state;
    uint32_t *stage[2];
};

__host__ __device__ inline int align4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline int lds_table_dwords(int S, int n_pots, int n_counters) {
    return align4(2 * S + 32 + n_pots + n_counters);
}

__device__ __forceinline__ LdsView carve(uint32_t *base, const StepArgs &a) {
    LdsView v;
    v.terrain = base;
    v.out_tab = base + a.S;
    v.rvalues = reinterpret_cast<int32_t *>(base + 2 * a.S);
    v.rtimes = v.rvalues + 16;
    v.pots = base + 2 * a.S + 32;
    v.counters = v.pots + a.n_pots;
    uint32_t *p = base + lds_table_dwords(a.S, a.n_pots, a.n_counters);
    v.state = p;
    p += align4(a.kpw * a.lstride);
    v.stage[0] = p;
    v.stage[1] = p + a.gpasses * kWave * kTileDwords;
    return v;
}

__device__ __forceinline__ int cook_time(const LdsView &L, const StepArgs &a, uint32_t obj) {
    if (a.uniform_time >= 0) return a.uniform_time;
    return L.rtimes[(4 * ((obj >> 8) & 0xff) + ((obj >> 16) & 0xff)) & 15];   // ref:56-57,156-157
}

__device__ __forceinline__ int dir_delta(int d, int W) {   // ref:22-32 move_in_direction
    return d == A_NORTH ? -W : d == A_SOUTH ? W : d == A_EAST ? 1 : d == A_WEST ? -1 : 0;
}

// ref:301-354 resolve_interacts, body of the per-player loop for one interacting player.
// `rec` is this kitchen's record in LDS; `held` the acting player's hands, `held_other` the
// other player's hands *as already mutated* by earlier players this step.
__device__ __forceinline__ int interact(uint32_t *rec, const LdsView &L, const StepArgs &a, int pos, int orient,
                                        uint32_t &held, uint32_t held_other, int pot_states) {
    int rew = 0;
    const int ipos = pos + dir_delta(orient, a.W);
    const uint32_t terr = L.terrain[ipos];
    const uint32_t hname = held & 0xff;
    if (terr == T_COUNTER) {                                              // ref:313-319
        const uint32_t t = rec[ipos];
        if (hname != O_NONE && (t & 0xff) == O_NONE) { rec[ipos] = held; held = 0; }
        else if (hname == O_NONE && (t & 0xff) != O_NONE) { held = t; rec[ipos] = 0; }
    } else if (terr == T_ONION_SRC) {                                     // ref:320-321
        if (hname == O_NONE) held = kOnionObj;
    } else if (terr == T_TOMATO_SRC) {                                    // ref:322-323
        if (hname == O_NONE) held = kTomatoObj;
    } else if (terr == T_DISH_SRC) {                                      // ref:324-327
        if (hname == O_NONE) {
            // ref:261-270 is_dish_pickup_useful: dishes in hands (acting player holds nothing) <
            // non-empty pots, and no dish lying on any counter.
            const int dishes = ((held_other & 0xff) == O_DISH) ? 1 : 0;
            if (dishes < pot_states) {
                bool on_counter = false;
                for (int i = 0; i < a.n_counters; ++i) on_counter |= ((rec[L.counters[i]] & 0xff) == O_DISH);
                if (!on_counter) rew += a.dish_rew;
            }
            held = kDishObj;
        }
    } else if (terr == T_POT && hname != O_NONE) {                        // ref:331-349
        uint32_t t = rec[ipos];
        int tick = static_cast<int32_t>(t) >> 24;
        const bool ready = (t & 0xff) != O_NONE && tick >= 0 && tick >= cook_time(L, a, t);   // ref:298-299
        if (hname == O_DISH && ready) {
            held = t; rec[ipos] = 0; rew += a.soup_rew;
        } else if (hname == O_ONION || hname == O_TOMATO) {
            if ((t & 0xff) == O_NONE) { t = kEmptySoup; tick = -1; }
            int n = ((t >> 8) & 0xff) + ((t >> 16) & 0xff);
            if (!(tick >= 0 || n == 3)) {
                t += (hname == O_ONION) ? 0x100u : 0x10000u;
                ++n; held = 0; rew += a.placement_rew;
            }
            // ref:287-296 + :348: idle (tick < 0) soup with exactly 3 ingredients starts cooking
            if ((t & 0xff) == O_SOUP && tick < 0 && n > 0 && n == 3) t &= 0x00FFFFFFu;
            rec[ipos] = t;
        }
    } else if (terr == T_SERVING && hname == O_SOUP) {                    // ref:350-353, 283-285
        rew += L.rvalues[(4 * ((held >> 8) & 0xff) + ((held >> 16) & 0xff)) & 15];
        held = 0;
    }
    return rew;
}

template <bool ACT64>
__device__ __forceinline__ int load_action(const void *actions, int64_t idx) {
    int a;
    if (ACT64) a = static_cast<int>(reinterpret_cast<const long long *>(actions)[idx]);
    else a = reinterpret_cast<const int *>(actions)[idx];
    return (static_cast<unsigned>(a) > 5u) ? A_STAY : a;   // out-of-range codes behave as STAY
}

__device__ __forceinline__ void store16(int8_t *dst, uint4 v) { *reinterpret_cast<uint4 *>(dst) = v; }
__device__ __forceinline__ void store4(int8_t *dst, uint32_t v) { *reinterpret_cast<uint32_t *>(dst) = v; }

// One launch = one env step for every kitchen (STEP) or a pure featurisation of the given
// state block (!STEP).  VEC16: every viewer plane base is 16-byte aligned.
template <bool STEP, bool ACT64, bool VEC16>
__global__ __launch_bounds__(kWave) void oc_step_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const LdsView L = carve(lds, a);
    const int lane = threadIdx.x;
    const int64_t k0 = static_cast<int64_t>(blockIdx.x) * a.kpw;
    const int n_valid = static_cast<int>(min(static_cast<int64_t>(a.kpw), a.n_kitchens - k0));
    const int S = a.S;

    // ---- tables -> LDS
    for (int i = lane; i < S; i += kWave) { L.terrain[i] = a.tabs->terrain[i]; L.out_tab[i] = a.tabs->out_tab[i]; }
    if (lane < 16) { L.rvalues[lane] = a.tabs->recipe_values[lane]; L.rtimes[lane] = a.tabs->recipe_times[lane]; }
    if (STEP) {
        if (lane < a.n_pots) L.pots[lane] = a.tabs->pots[lane];
        for (int i = lane; i < a.n_counters; i += kWave) L.counters[i] = a.tabs->counters[i];
    }

    // ---- state block -> LDS (16 B per lane per load, records are whole uint4s)
    {
        const uint4 *src = a.state + k0 * a.recq;
        const int nq = n_valid * a.recq;
        for (int c = lane; c < nq; c += kWave) {
            const uint4 q = src[c];
            const int k = c / a.recq, r = c - k * a.recq;
            uint32_t *d = L.state + k * a.lstride + 4 * r;
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        }
    }
    __syncthreads();

    // ---- phase T: lane = kitchen.  ref:381-385 get_state_transition + wrappers' done/reset rule
    if (STEP && lane < n_valid) {
        uint32_t *rec = L.state + lane * a.lstride;
        const int64_t kid = k0 + lane;
        int a0 = load_action<ACT64>(a.actions, kid);
        int a1 = load_action<ACT64>(a.actions, a.n_kitchens + kid);
        uint32_t posor = rec[S], h0 = rec[S + 1], h1 = rec[S + 2];
        int ts = static_cast<int>(rec[S + 3]);
        int pos0 = posor & 0xff, or0 = (posor >> 8) & 0xff, pos1 = (posor >> 16) & 0xff, or1 = posor >> 24;

        int rew = 0;
        if (a0 == A_INTERACT || a1 == A_INTERACT) {
            // ref:272-281 get_pot_states, taken once before anybody acts (ref:302)
            int pot_states = 0;
            for (int i = 0; i < a.n_pots; ++i) {
                const uint32_t o = rec[L.pots[i]];
                if ((o & 0xff) != O_NONE) {
                    const int tick = static_cast<int32_t>(o) >> 24;
                    const int n = ((o >> 8) & 0xff) + ((o >> 16) & 0xff);
                    pot_states += (tick >= 0 || n < 3) ? 1 : 0;
                }
            }
            if (a0 == A_INTERACT) rew += interact(rec, L, a, pos0, or0, h0, h1, pot_states);
            if (a1 == A_INTERACT) rew += interact(rec, L, a, pos1, or1, h1, h0, pot_states);
        }

        // ref:393-399 _move_if_direction, ref:356-366 _handle_collisions
        int np0 = pos0, no0 = or0, np1 = pos1, no1 = or1;
        if (a0 != A_INTERACT) {
            const int cand = pos0 + dir_delta(a0, a.W);
            no0 = (a0 == A_STAY) ? or0 : a0;
            np0 = (L.terrain[cand] != T_AIR) ? pos0 : cand;
        }
        if (a1 != A_INTERACT) {
            const int cand = pos1 + dir_delta(a1, a.W);
            no1 = (a1 == A_STAY) ? or1 : a1;
            np1 = (L.terrain[cand] != T_AIR) ? pos1 : cand;
        }
        const bool blocked = (np0 == np1) || (np0 == pos1 && pos0 == np1);
        if (!blocked) { pos0 = np0; pos1 = np1; }
        or0 = no0; or1 = no1;

        // ref:373-375 timestep; envs/overcooked2_env.py:334 done; vectorenv.py:369-370 reset rule
        ts += 1;
        const int is_done = ts >= a.horizon;
        posor = pos0 | (or0 << 8) | (pos1 << 16) | (static_cast<uint32_t>(or1) << 24);
        if (is_done) { posor = a.start_posor; h0 = 0; h1 = 0; ts = 0; }
        rec[S] = posor; rec[S + 1] = h0; rec[S + 2] = h1; rec[S + 3] = static_cast<uint32_t>(ts);
        rec[4 * a.recq] = is_done;                       // flag word (the odd-stride pad dword)
        a.reward[kid] = rew;                             // envs/overcooked2_env.py:336-339: sum, to every player
        a.reward[a.n_kitchens + kid] = rew;
        a.done[kid] = is_done;
    }
    if (STEP) __syncthreads();

    // ---- phase E: lane = tile, in output order o = x*H + y per kitchen (envs/overcooked2_env.py:324)
    const int ntiles = n_valid * S;
    const int npass = (ntiles + kWave - 1) / kWave;
    int k = lane / S, o = lane - k * S;
    const int dk = kWave / S, dq = kWave - dk * S;
    const int64_t plane_base = k0 * S * kChannels;       // byte offset of this wave's kitchens in a viewer plane
    for (int p0 = 0; p0 < npass; p0 += a.gpasses) {
        const int pend = min(p0 + a.gpasses, npass);
        for (int p = p0; p < pend; ++p) {
            const int f = p * kWave + lane;
            if (f < ntiles) {
                uint32_t *rec = L.state + k * a.lstride;
                const uint32_t tab = L.out_tab[o];
                const int t = tab & 0xff;
                const uint32_t terr = tab >> 8;
                uint32_t obj = rec[t];
                if (STEP) {
                    // ref:376-379: every SOUP lying on a tile that is cooking ticks (held ones do not)
                    const int tick = static_cast<int32_t>(obj) >> 24;
                    if ((obj & 0xff) == O_SOUP && tick >= 0 && tick < cook_time(L, a, obj)) obj += 0x01000000u;
                    if (rec[4 * a.recq]) obj = 0;        // kitchen was reset this step
                    rec[t] = obj;
                }
                const uint32_t posor = rec[S], h0 = rec[S + 1] & 0xff, h1 = rec[S + 2] & 0xff;
                const bool here0 = (posor & 0xff) == static_cast<uint32_t>(t);
                const bool here1 = ((posor >> 16) & 0xff) == static_cast<uint32_t>(t);
                // ref:221-234: 4 one-hot orientation bytes per player, placed at the player's tile
                const uint32_t A = here0 ? (1u << (8 * ((posor >> 8) & 3))) : 0u;
                const uint32_t B = here1 ? (1u << (8 * ((posor >> 24) & 3))) : 0u;
                const uint32_t e0 = here0 ? 1u : 0u, e1 = here1 ? 1u : 0u;
                const uint32_t name = obj & 0xff;
                const bool inpot = (name == O_SOUP) && (terr == T_POT);
                const int tk = static_cast<int32_t>(obj) >> 24;
                // ref:165-171 static terrain one-hot at channel 10 + terrain - 1
                const uint32_t terr2 = terr == T_POT ? 0x00010000u : terr == T_COUNTER ? 0x01000000u : 0u;
                const uint32_t terr3 = terr >= T_ONION_SRC ? (1u << (8 * (terr - T_ONION_SRC))) : 0u;
                // ref:182-211 objects on tiles, ref:236-253 objects in hands
                const uint32_t soupf = ((name == O_SOUP && !inpot) || (here0 && h0 == O_SOUP) || (here1 && h1 == O_SOUP)) ? 1u : 0u;
                const uint32_t dishf = ((name == O_DISH) || (here0 && h0 == O_DISH) || (here1 && h1 == O_DISH)) ? 1u : 0u;
                const uint32_t onionf = ((name == O_ONION) || (here0 && h0 == O_ONION) || (here1 && h1 == O_ONION)) ? 1u : 0u;
                const uint32_t d3 = inpot ? ((terr3 & 0x00FFFFFFu) | (((obj >> 8) & 0xff) << 24)) : terr3;
                const uint32_t d4 = (inpot && tk > 0 ? static_cast<uint32_t>(tk) : 0u) | (soupf << 8) | (dishf << 16) | (onionf << 24);
                const int slot = (f - p0 * kWave) * kTileDwords;
                uint32_t *s0 = L.stage[0] + slot, *s1 = L.stage[1] + slot;
                s0[0] = e0 | (e1 << 8) | (A << 16); s0[1] = (A >> 16) | (B << 16); s0[2] = (B >> 16) | terr2; s0[3] = d3; s0[4] = d4;
                s1[0] = e1 | (e0 << 8) | (B << 16); s1[1] = (B >> 16) | (A << 16); s1[2] = (A >> 16) | terr2; s1[3] = d3; s1[4] = d4;
            }
            k += dk; o += dq;
            if (o >= S) { o -= S; ++k; }
        }
        __syncthreads();
        // ---- flush the staged tiles of both viewers: contiguous bytes in the output planes
        const int gbytes = (min(ntiles - p0 * kWave, a.gpasses * kWave)) * kChannels;
        const int64_t goff = plane_base + static_cast<int64_t>(p0) * kWave * kChannels;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            int8_t *dst = a.obs + v * a.obs_player_stride + goff;
            const uint32_t *src = L.stage[v];
            if (VEC16) {
                for (int c = lane; c * 16 < gbytes; c += kWave) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(src + 4 * c);
                    if (c * 16 + 16 <= gbytes) store16(dst + c * 16, q);
                    else {                                    // ragged tail of the very last wave only
                        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                        for (int j = 0; j < 4; ++j) if (c * 16 + 4 * j < gbytes) store4(dst + c * 16 + 4 * j, w[j]);
                    }
                }
            } else {
                for (int c = lane; c * 4 < gbytes; c += kWave) store4(dst + c * 4, src[c]);
            }
        }
        __syncthreads();
    }

    // ---- state block LDS -> HBM
    if (STEP) {
        uint4 *dst = a.state + k0 * a.recq;
        const int nq = n_valid * a.recq;
        for (int c = lane; c < nq; c += kWave) {
            const int kk = c / a.recq, r = c - kk * a.recq;
            const uint32_t *s = L.state + kk * a.lstride + 4 * r;
            dst[c] = make_uint4(s[0], s[1], s[2], s[3]);
        }
    }
}

// every kitchen to the standard start state (ref:387-391): empty tiles, players at their start
// tiles facing NORTH with empty hands, timestep 0.
__global__ void oc_fill_start_kernel(uint4 *state, int64_t first, int64_t count, int recq, int S, uint32_t start_posor) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= count * recq) return;
    const int r = static_cast<int>(c % recq);
    uint32_t w[4] = {0, 0, 0, 0};
    const int idx = S - 4 * r;                 // dword S holds posor
    if (idx >= 0 && idx < 4) w[idx] = start_posor;
    state[first * recq + c] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------------
// host side
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define OC_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t err__ = (call);                                                                \
        if (err__ != hipSuccess)                                                                  \
            return fail(OC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct oc_engine {
    oc_layout_params params;
    int64_t n;
    int device;
    int S, recq;
    int n_pots, n_counters, uniform_time;
    uint32_t start_posor;
    int kpw, gpasses;
    uint4 *d_state;
    DevTables *d_tabs;
    hipEvent_t ev0, ev1;
};

namespace {

size_t lds_bytes(const oc_engine *e, int kpw, int gpasses) {
    const int lstride = 4 * e->recq + 1;
    return sizeof(uint32_t) * (static_cast<size_t>(lds_table_dwords(e->S, e->n_pots, e->n_counters)) +
                               align4(kpw * lstride) + 2 * gpasses * kWave * kTileDwords);
}

StepArgs make_args(const oc_engine *e, uint4 *state, int64_t n, const void *actions, int8_t *obs, int32_t *reward,
                   int32_t *done) {
    StepArgs a;
    a.state = state;
    a.tabs = e->d_tabs;
    a.actions = actions;
    a.obs = obs;
    a.reward = reward;
    a.done = done;
    a.n_kitchens = n;
    a.obs_player_stride = n * e->S * kChannels;
    a.W = e->params.width;
    a.S = e->S;
    a.recq = e->recq;
    a.lstride = 4 * e->recq + 1;
    a.kpw = e->kpw;
    a.gpasses = e->gpasses;
    a.horizon = e->params.horizon;
    a.placement_rew = e->params.placement_in_pot_rew;
    a.dish_rew = e->params.dish_pickup_rew;
    a.soup_rew = e->params.soup_pickup_rew;
    a.n_pots = e->n_pots;
    a.n_counters = e->n_counters;
    a.uniform_time = e->uniform_time;
    a.start_posor = e->start_posor;
    return a;
}

template <bool STEP, bool ACT64>
int launch(const oc_engine *e, const StepArgs &a, hipStream_t s) {
    if (a.n_kitchens == 0) return OC_OK;
    const bool vec16 = (reinterpret_cast<uintptr_t>(a.obs) % 16 == 0) && (a.obs_player_stride % 16 == 0);
    const unsigned grid = static_cast<unsigned>((a.n_kitchens + a.kpw - 1) / a.kpw);
    const size_t shmem = lds_bytes(e, a.kpw, a.gpasses);
    if (vec16) hipLaunchKernelGGL((oc_step_kernel<STEP, ACT64, true>), dim3(grid), dim3(kWave), shmem, s, a);
    else hipLaunchKernelGGL((oc_step_kernel<STEP, ACT64, false>), dim3(grid), dim3(kWave), shmem, s, a);
    OC_HIP(hipGetLastError());
    return OC_OK;
}

int check_engine(const oc_engine *e) {
    if (!e) return fail(OC_ERR_INVALID_ARG, "engine handle is NULL");
    return OC_OK;
}

int step_impl(oc_engine *e, const void *d_actions, int action_dtype, int8_t *d_obs, int32_t *d_reward,
              int32_t *d_done, hipStream_t s) {
    const StepArgs a = make_args(e, e->d_state, e->n, d_actions, d_obs, d_reward, d_done);
    return action_dtype == OC_ACT_I64 ? launch<true, true>(e, a, s) : launch<true, false>(e, a, s);
}

}  // namespace

extern "C" {

const char *oc_last_error(void) { return g_err; }
int oc_abi_version(void) { return OC_ABI_VERSION; }

int oc_create(const oc_layout_params *p, int64_t n_kitchens, int device_id, oc_engine **out) {
    if (!p || !out) return fail(OC_ERR_INVALID_ARG, "oc_create: NULL argument");
    *out = nullptr;
    if (n_kitchens < 1) return fail(OC_ERR_INVALID_ARG, "oc_create: n_kitchens must be >= 1 (got %lld)", (long long)n_kitchens);
    if (p->num_players != 2) return fail(OC_ERR_UNSUPPORTED, "oc_create: only 2-player kitchens are supported (got %d)", p->num_players);
    if (p->width < 1 || p->height < 1 || p->width * p->height > OC_MAX_TILES)
        return fail(OC_ERR_UNSUPPORTED, "oc_create: grid %dx%d outside 1..%d tiles", p->width, p->height, OC_MAX_TILES);
    const int S = p->width * p->height;
    if (p->horizon < 1) return fail(OC_ERR_INVALID_ARG, "oc_create: horizon must be >= 1");

    std::vector<DevTables> tabs_vec(1);
    DevTables &tabs = tabs_vec[0];
    memset(&tabs, 0, sizeof(tabs));
    int n_pots = 0, n_counters = 0;
    for (int t = 0; t < S; ++t) {
        const int terr = p->terrain[t];
        if (terr < 0 || terr > 6) return fail(OC_ERR_INVALID_ARG, "oc_create: terrain[%d] = %d outside 0..6", t, terr);
        const int x = t % p->width, y = t / p->width;
        // the reference relies on the border being non-AIR so that moves never leave the grid
        if ((x == 0 || y == 0 || x == p->width - 1 || y == p->height - 1) && terr == T_AIR)
            return fail(OC_ERR_INVALID_ARG, "oc_create: border tile (%d,%d) is walkable; layouts must be walled", x, y);
        tabs.terrain[t] = terr;
        tabs.out_tab[x * p->height + y] = t | (terr << 8);
        if (terr == T_POT) {
            if (n_pots == kMaxPots) return fail(OC_ERR_UNSUPPORTED, "oc_create: more than %d pots", kMaxPots);
            tabs.pots[n_pots++] = t;
        }
        if (terr == T_COUNTER) tabs.counters[n_counters++] = t;
    }
    int uniform_time = p->recipe_times[0];
    for (int i = 0; i < 16; ++i) {
        if (p->recipe_times[i] < 0 || p->recipe_times[i] > 127)
            return fail(OC_ERR_UNSUPPORTED, "oc_create: recipe_times[%d] = %d outside 0..127 (tick is a signed byte)", i, p->recipe_times[i]);
        if (p->recipe_times[i] != p->recipe_times[0]) uniform_time = -1;
        tabs.recipe_times[i] = p->recipe_times[i];
        tabs.recipe_values[i] = p->recipe_values[i];
    }
    uint32_t start_posor = 0;
    for (int i = 0; i < 2; ++i) {
        const int x = p->start_player_x[i], y = p->start_player_y[i];
        if (x < 0 || y < 0 || x >= p->width || y >= p->height || p->terrain[y * p->width + x] != T_AIR)
            return fail(OC_ERR_INVALID_ARG, "oc_create: start position of player %d (%d,%d) is not a free tile", i, x, y);
        start_posor |= static_cast<uint32_t>(y * p->width + x) << (16 * i);
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(OC_ERR_NO_DEVICE, "oc_create: no HIP device visible (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(OC_ERR_INVALID_ARG, "oc_create: device_id %d outside 0..%d", device_id, ndev - 1);
    OC_HIP(hipSetDevice(device_id));

    oc_engine *e = new (std::nothrow) oc_engine();
    if (!e) return fail(OC_ERR_INVALID_ARG, "oc_create: out of host memory");
    e->params = *p;
    e->n = n_kitchens;
    e->device = device_id;
    e->S = S;
    e->recq = (S + 4 + 3) / 4;
    e->n_pots = n_pots;
    e->n_counters = n_counters;
    e->uniform_time = uniform_time;
    e->start_posor = start_posor;
    e->kpw = 16;
    e->gpasses = 4;
    e->d_state = nullptr;
    e->d_tabs = nullptr;
    e->ev0 = e->ev1 = nullptr;
    hipError_t err = hipMalloc(reinterpret_cast<void **>(&e->d_state), static_cast<size_t>(n_kitchens) * e->recq * sizeof(uint4));
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void **>(&e->d_tabs), sizeof(DevTables));
    if (err == hipSuccess) err = hipMemcpy(e->d_tabs, &tabs, sizeof(DevTables), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipEventCreate(&e->ev0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev1);
    if (err != hipSuccess) {
        fail(OC_ERR_HIP, "oc_create: device allocation failed: %s", hipGetErrorString(err));
        oc_destroy(e);
        return OC_ERR_HIP;
    }
    int rc = oc_reset_all(e, nullptr, nullptr);
    if (rc == OC_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(OC_ERR_HIP, "oc_create: start-state fill failed");
    if (rc != OC_OK) { oc_destroy(e); return rc; }
    *out = e;
    return OC_OK;
}

int oc_destroy(oc_engine *e) {
    if (!e) return OC_OK;
    (void)hipSetDevice(e->device);
    if (e->d_state) (void)hipFree(e->d_state);
    if (e->d_tabs) (void)hipFree(e->d_tabs);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return OC_OK;
}

int oc_step(oc_engine *e, const void *d_actions, int action_dtype, int8_t *d_obs, int32_t *d_reward, int32_t *d_done,
            void *stream) {
    if (int rc = check_engine(e)) return rc;
    if (!d_actions || !d_obs || !d_reward || !d_done) return fail(OC_ERR_INVALID_ARG, "oc_step: NULL device buffer");
    if (action_dtype != OC_ACT_I32 && action_dtype != OC_ACT_I64) return fail(OC_ERR_INVALID_ARG, "oc_step: bad action dtype %d", action_dtype);
    if (reinterpret_cast<uintptr_t>(d_obs) % 4) return fail(OC_ERR_INVALID_ARG, "oc_step: obs buffer must be 4-byte aligned");
    return step_impl(e, d_actions, action_dtype, d_obs, d_reward, d_done, static_cast<hipStream_t>(stream));
}

int oc_observe(oc_engine *e, int8_t *d_obs, void *stream) {
    if (int rc = check_engine(e)) return rc;
    if (!d_obs) return fail(OC_ERR_INVALID_ARG, "oc_observe: NULL obs buffer");
    if (reinterpret_cast<uintptr_t>(d_obs) % 4) return fail(OC_ERR_INVALID_ARG, "oc_observe: obs buffer must be 4-byte aligned");
    const StepArgs a = make_args(e, e->d_state, e->n, nullptr, d_obs, nullptr, nullptr);
    return launch<false, false>(e, a, static_cast<hipStream_t>(stream));
}

int oc_encode_states(oc_engine *e, const void *d_states, int64_t n, int8_t *d_obs, void *stream) {
    if (int rc = check_engine(e)) return rc;
    if (!d_states || !d_obs || n < 0) return fail(OC_ERR_INVALID_ARG, "oc_encode_states: bad argument");
    if (reinterpret_cast<uintptr_t>(d_states) % 16) return fail(OC_ERR_INVALID_ARG, "oc_encode_states: state block must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_obs) % 4) return fail(OC_ERR_INVALID_ARG, "oc_encode_states: obs buffer must be 4-byte aligned");
    const StepArgs a = make_args(e, const_cast<uint4 *>(static_cast<const uint4 *>(d_states)), n, nullptr, d_obs, nullptr, nullptr);
    return launch<false, false>(e, a, static_cast<hipStream_t>(stream));
}

int oc_reset_all(oc_engine *e, int8_t *d_obs, void *stream) {
    if (int rc = check_engine(e)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t chunks = e->n * e->recq;
    const unsigned grid = static_cast<unsigned>((chunks + 255) / 256);
    hipLaunchKernelGGL(oc_fill_start_kernel, dim3(grid), dim3(256), 0, s, e->d_state, (int64_t)0, e->n, e->recq, e->S, e->start_posor);
    OC_HIP(hipGetLastError());
    if (d_obs) return oc_observe(e, d_obs, stream);
    return OC_OK;
}

int oc_rollout(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T, int8_t *d_obs,
               int64_t obs_step_stride, int32_t *d_reward, int64_t reward_step_stride, int32_t *d_done,
               int64_t done_step_stride, void *stream) {
    if (int rc = check_engine(e)) return rc;
    if (!d_actions || !d_obs || !d_reward || !d_done || T_act < 1 || T < 0)
        return fail(OC_ERR_INVALID_ARG, "oc_rollout: bad argument");
    if (action_dtype != OC_ACT_I32 && action_dtype != OC_ACT_I64) return fail(OC_ERR_INVALID_ARG, "oc_rollout: bad action dtype %d", action_dtype);
    if (reinterpret_cast<uintptr_t>(d_obs) % 4 || obs_step_stride % 4) return fail(OC_ERR_INVALID_ARG, "oc_rollout: obs buffer/stride must be 4-byte aligned");
    const size_t asz = action_dtype == OC_ACT_I64 ? 8 : 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int64_t t = 0; t < T; ++t) {
        const char *act = static_cast<const char *>(d_actions) + static_cast<size_t>(t % T_act) * 2 * e->n * asz;
        if (int rc = step_impl(e, act, action_dtype, d_obs + t * obs_step_stride, d_reward + t * reward_step_stride,
                               d_done + t * done_step_stride, s))
            return rc;
    }
    return OC_OK;
}

int oc_rollout_timed(oc_engine *e, const void *d_actions, int action_dtype, int64_t T_act, int64_t T, int8_t *d_obs,
                     int32_t *d_reward, int32_t *d_done, void *stream, float *elapsed_ms) {
    if (int rc = check_engine(e)) return rc;
    if (!elapsed_ms) return fail(OC_ERR_INVALID_ARG, "oc_rollout_timed: NULL elapsed_ms");
    hipStream_t s = static_cast<hipStream_t>(stream);
    OC_HIP(hipEventRecord(e->ev0, s));
    if (int rc = oc_rollout(e, d_actions, action_dtype, T_act, T, d_obs, 0, d_reward, 0, d_done, 0, stream)) return rc;
    OC_HIP(hipEventRecord(e->ev1, s));
    OC_HIP(hipEventSynchronize(e->ev1));
    OC_HIP(hipEventElapsedTime(elapsed_ms, e->ev0, e->ev1));
    return OC_OK;
}

int64_t oc_state_record_bytes(const oc_engine *e) { return e ? static_cast<int64_t>(e->recq) * 16 : 0; }
int64_t oc_num_kitchens(const oc_engine *e) { return e ? e->n : 0; }
int oc_device_id(const oc_engine *e) { return e ? e->device : -1; }

int oc_obs_shape(const oc_engine *e, int64_t shape5[5]) {
    if (int rc = check_engine(e)) return rc;
    if (!shape5) return fail(OC_ERR_INVALID_ARG, "oc_obs_shape: NULL output");
    shape5[0] = 2; shape5[1] = e->n; shape5[2] = e->params.width; shape5[3] = e->params.height; shape5[4] = kChannels;
    return OC_OK;
}

static int check_range(const oc_engine *e, int64_t first, int64_t count, const char *who) {
    if (first < 0 || count < 0 || first + count > e->n)
        return fail(OC_ERR_INVALID_ARG, "%s: range [%lld, %lld) outside 0..%lld", who, (long long)first, (long long)(first + count), (long long)e->n);
    return OC_OK;
}

int oc_get_state(oc_engine *e, int64_t first, int64_t count, void *host_dst) {
    if (int rc = check_engine(e)) return rc;
    if (!host_dst) return fail(OC_ERR_INVALID_ARG, "oc_get_state: NULL destination");
    if (int rc = check_range(e, first, count, "oc_get_state")) return rc;
    OC_HIP(hipSetDevice(e->device));
    OC_HIP(hipDeviceSynchronize());
    OC_HIP(hipMemcpy(host_dst, e->d_state + first * e->recq, static_cast<size_t>(count) * e->recq * 16, hipMemcpyDeviceToHost));
    return OC_OK;
}

int oc_set_state(oc_engine *e, int64_t first, int64_t count, const void *host_src) {
    if (int rc = check_engine(e)) return rc;
    if (!host_src) return fail(OC_ERR_INVALID_ARG, "oc_set_state: NULL source");
    if (int rc = check_range(e, first, count, "oc_set_state")) return rc;
    // reject records the kernels could fault on: positions must index real tiles
    const uint32_t *w = static_cast<const uint32_t *>(host_src);
    for (int64_t k = 0; k < count; ++k) {
        const uint32_t posor = w[k * e->recq * 4 + e->S];
        const int p0 = posor & 0xff, o0 = (posor >> 8) & 0xff, p1 = (posor >> 16) & 0xff, o1 = posor >> 24;
        const int W = e->params.width, H = e->params.height;
        auto interior = [&](int p) { const int x = p % W, y = p / W; return p < e->S && x > 0 && y > 0 && x < W - 1 && y < H - 1; };
        if (!interior(p0) || !interior(p1) || o0 > 3 || o1 > 3)
            return fail(OC_ERR_INVALID_ARG, "oc_set_state: record %lld has a player off the interior grid or a bad orientation", (long long)(first + k));
    }
    OC_HIP(hipSetDevice(e->device));
    OC_HIP(hipDeviceSynchronize());
    OC_HIP(hipMemcpy(e->d_state + first * e->recq, host_src, static_cast<size_t>(count) * e->recq * 16, hipMemcpyHostToDevice));
    return OC_OK;
}

int oc_export_states(oc_engine *e, int64_t first, int64_t count, void *d_dst, void *stream) {
    if (int rc = check_engine(e)) return rc;
    if (!d_dst) return fail(OC_ERR_INVALID_ARG, "oc_export_states: NULL destination");
    if (int rc = check_range(e, first, count, "oc_export_states")) return rc;
    OC_HIP(hipMemcpyAsync(d_dst, e->d_state + first * e->recq, static_cast<size_t>(count) * e->recq * 16, hipMemcpyDeviceToDevice,
                          static_cast<hipStream_t>(stream)));
    return OC_OK;
}

int oc_import_states(oc_engine *e, int64_t first, int64_t count, const void *d_src, void *stream) {
    if (int rc = check_engine(e)) return rc;
    if (!d_src) return fail(OC_ERR_INVALID_ARG, "oc_import_states: NULL source");
    if (int rc = check_range(e, first, count, "oc_import_states")) return rc;
    OC_HIP(hipMemcpyAsync(e->d_state + first * e->recq, d_src, static_cast<size_t>(count) * e->recq * 16, hipMemcpyDeviceToDevice,
                          static_cast<hipStream_t>(stream)));
    return OC_OK;
}

int oc_set_tuning(oc_engine *e, int kitchens_per_wave, int passes_per_flush) {
    if (int rc = check_engine(e)) return rc;
    const int kpw = kitchens_per_wave ? kitchens_per_wave : e->kpw;
    const int gp = passes_per_flush ? passes_per_flush : e->gpasses;
    if (kpw < 4 || kpw > 64 || kpw % 4) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: kitchens_per_wave must be a multiple of 4 in 4..64");
    if (gp < 1 || gp > 8) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: passes_per_flush must be in 1..8");
    if (lds_bytes(e, kpw, gp) > 64 * 1024) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: %zu B of LDS per wave exceeds 64 KiB", lds_bytes(e, kpw, gp));
    e->kpw = kpw;
    e->gpasses = gp;
    return OC_OK;
}

}  // extern "C"
