// overcooked_engine.hip -- MI355X (gfx950) Overcooked rollout engine: kernels + C ABI.
//
// What it replaces: the reference's Madrona simulation of the Overcooked MDP
// (/root/reference/src/overcooked2_env/sim.cpp: 13 task-graph nodes per step, :649-688) and its
// Manager/bindings (mgr.cpp, bindings.cpp).  Semantics follow the reference's *Python* twin
// (envs/overcooked2_reimplement.py, DummyMDP) which SURVEY.md 8(a) makes normative; reference
// line numbers below are of that file unless another file is named.
//
// Execution model (see DESIGN.md for the roofline argument):
//   * one 64-lane wavefront (= one workgroup) walks groups of KPW consecutive kitchens in a
//     grid-stride loop; the next group's state records and actions are prefetched into registers
//     while the current group is processed, and the current group's plane stores drain while the
//     next group is computed (without this every wave of the chip sits in the same phase);
//   * state records are flat-copied HBM -> LDS with 16-byte loads (odd LDS stride per kitchen so
//     lane-per-kitchen accesses are bank-conflict free);
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
#include <cstdlib>
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

// ---- INTERACT outcome table ---------------------------------------------------------------------
// ref:301-354.  Which branch of resolve_interacts fires depends only on the faced terrain (7), the
// name of the held object (5) and a 5-way class of the faced tile's object; the kernel looks the
// outcome up instead of re-deriving ~40 predicates per player per step.
//   tile class: 0 no object | 1 object with tick >= 0, not ready (cooking) | 2 tick < 0 and
//   onions+tomatoes != 3 (idle, has room) | 3 ready (tick >= 0 and tick >= cook time) | 4 tick < 0 and
//   exactly 3 ingredients (idle and full: starts cooking when poked with an ingredient, ref:348)
enum : uint32_t {
    F_HELD_GETS_TILE = 1u,      // held := faced object                      (counter pick-up, soup pick-up)
    F_HELD_CLEARED = 2u,        // held := nothing                           (counter drop, ingredient added, serve)
    F_HELD_FRESH = 4u,          // held := fresh object, name in bits 12..14 (dispensers)
    F_TILE_GETS_HELD = 8u,      // faced := held                             (counter drop)
    F_TILE_CLEARED = 16u,       // faced := nothing                          (counter pick-up, soup pick-up)
    F_ADD = 32u,                // faced soup (created if absent) gains the held ingredient
    F_START_CHECK = 64u,        // idle soup with exactly 3 ingredients starts cooking
    F_DISH_CHECK = 128u,        // dish pick-up: reward iff is_dish_pickup_useful
    F_SERVE = 256u,             // reward = recipe value of the served soup
    // bits 16..31: fixed reward of the outcome (soup pick-up / placement), filled per engine
};
constexpr int kTileClasses = 5;
constexpr int kLutSize = 7 * 5 * kTileClasses;

inline uint32_t interact_outcome(uint32_t terr, uint32_t hname, int tclass, uint32_t soup_pickup_rew, uint32_t placement_rew) {
    const bool holding = hname != O_NONE, occupied = tclass != 0;
    if (terr == T_COUNTER) {                                                   // ref:313-319
        if (holding && !occupied) return F_TILE_GETS_HELD | F_HELD_CLEARED;
        if (!holding && occupied) return F_HELD_GETS_TILE | F_TILE_CLEARED;
        return 0;
    }
    if (terr == T_ONION_SRC && !holding) return F_HELD_FRESH | (O_ONION << 12);  // ref:320-321
    if (terr == T_TOMATO_SRC && !holding) return F_HELD_FRESH | (O_TOMATO << 12); // ref:322-323
    if (terr == T_DISH_SRC && !holding) return F_HELD_FRESH | (O_DISH << 12) | F_DISH_CHECK;   // ref:324-327
    if (terr == T_POT && holding) {                                            // ref:331-349
        if (hname == O_DISH && tclass == 3) return F_HELD_GETS_TILE | F_TILE_CLEARED | (soup_pickup_rew << 16);
        if (hname == O_ONION || hname == O_TOMATO) {
            const bool room = tclass == 0 || tclass == 2;                      // not (tick >= 0 or n == 3)
            if (room) return F_ADD | F_HELD_CLEARED | F_START_CHECK | (placement_rew << 16);
            return tclass == 4 ? F_START_CHECK : 0u;
        }
        return 0;
    }
    if (terr == T_SERVING && hname == O_SOUP) return F_HELD_CLEARED | F_SERVE;   // ref:350-353
    return 0;
}

// Per-layout lookup tables, resident in HBM (L2-hot), copied to LDS by every wave.
struct DevTables {
    uint32_t terrain[256];      // by state tile index t = y*W + x
    uint32_t out_tab2[512];     // by output tile index o = x*H + y, two dwords:
                                //   [0] t | is_pot<<8 | (terrain bits of channels 10,11 as the high half of dword 2)<<16
                                //   [1] terrain bits of channels 12..15 (dword 3)
    int32_t recipe_values[16];  // ref:283-285
    int32_t recipe_times[16];   // ref:156-157
    uint32_t interact_lut[kLutSize];   // outcome of INTERACT by (terrain, held name, faced-tile class): build_interact_lut()
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
    int32_t n_groups;           // ceil(n_kitchens / kpw)
    int32_t horizon, placement_rew, dish_rew, soup_rew;
    int32_t n_pots, n_counters;
    int32_t pots4[4];           // first four POT tile indices (rest: DevTables::pots)
    unsigned long long delta_pack;  // signed bytes {-W, +W, +1, -1, 0, 0}: tile-index delta per direction/action
    int32_t uniform_time;       // recipe time if all 16 are equal, else -1
    uint32_t start_posor;       // p0.pos | NORTH<<8 | p1.pos<<16 | NORTH<<24   (ref:387-391)
    int32_t ablate;             // timing diagnostics only (env OC_ABLATE at oc_create); 0 in production
    unsigned long long *trace;  // timing diagnostics only: [wave][8] s_memrealtime stamps (100 MHz), else NULL
};

struct LdsView {
    uint32_t *terrain, *pots, *counters, *ilut;
    uint2 *out_tab2;
    int32_t *rvalues, *rtimes;
    uint32_t *state;
    uint32_t *kd;               // per-kitchen derived block, 8 dwords: pos0|pos1<<8|done<<16, oh0, oh1, hf0, hf1
    uint32_t *stage[2];
};

constexpr int kMaxQ = 5;        // prefetch registers: up to 5 uint4 per lane => kpw * recq <= 320 (template QN <= kMaxQ)
constexpr int kKd = 8;

__host__ __device__ inline int align4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline int lds_table_dwords(int S, int n_pots, int n_counters) {
    return align4(3 * S + 32 + kLutSize + n_pots + n_counters);
}

__device__ __forceinline__ LdsView carve(uint32_t *base, const StepArgs &a) {
    LdsView v;
    v.out_tab2 = reinterpret_cast<uint2 *>(base);                 // 8-byte aligned pairs
    v.terrain = base + 2 * a.S;
    v.rvalues = reinterpret_cast<int32_t *>(base + 3 * a.S);
    v.rtimes = v.rvalues + 16;
    v.ilut = base + 3 * a.S + 32;
    v.pots = v.ilut + kLutSize;
    v.counters = v.pots + a.n_pots;
    uint32_t *p = base + lds_table_dwords(a.S, a.n_pots, a.n_counters);
    v.kd = p;                                                      // 16-byte aligned rows
    p += a.kpw * kKd;
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

// ref:22-32 move_in_direction: tile-index delta of direction/action d in 0..5, looked up in six signed
// bytes {-W, +W, +1, -1, 0, 0} (a select chain here is lowered to a divergent branch tree by the compiler)
__device__ __forceinline__ int dir_delta(int d, unsigned long long pack) {
    return static_cast<int>(static_cast<signed char>(pack >> (8 * d)));
}

// ref:301-354 resolve_interacts, body of the per-player loop for one player: table look-up of the
// branch that fires, then branch-free application of its effects.
//   act        this player chose INTERACT
//   terr, t    terrain and object of the faced tile (t as it stands when this player acts)
//   held       the acting player's hands (in/out); held_other: the other player's hands as already
//              mutated by earlier players this step;  rv = recipe value of `held` were it served
// Returns the reward; t is updated in place (caller writes it back to the faced tile).
__device__ __forceinline__ int interact(bool act, uint32_t terr, uint32_t &t, uint32_t &held, uint32_t held_other,
                                        int rv, int pot_states, const uint32_t *rec, const LdsView &L,
                                        const StepArgs &a) {
    const uint32_t hname = held & 0xff;
    const int tick = static_cast<int32_t>(t) >> 24;
    const int n = static_cast<int>(((t >> 8) & 0xff) + ((t >> 16) & 0xff));
    const bool idle = tick < 0;
    const bool upper = idle ? (n == 3) : (tick >= cook_time(L, a, t));       // class 4 / class 3
    const int tclass = ((t & 0xff) != O_NONE ? 1 : 0) * (1 + (idle ? 1 : 0) + (upper ? 2 : 0));
    uint32_t out = L.ilut[(terr * 5 + hname) * kTileClasses + tclass];
    out = act ? out : 0u;
    int rew = 0;
    // ref:261-270 is_dish_pickup_useful: dishes in hands (the acting player holds nothing) < non-empty
    // pots and no dish lying on any counter -- the scan runs only for lanes that need its answer
    if ((out & F_DISH_CHECK) && (((held_other & 0xff) == O_DISH) ? 1 : 0) < pot_states) {
        bool lying = false;
        for (int i = 0; i < a.n_counters; ++i) lying |= ((rec[L.counters[i]] & 0xff) == O_DISH);
        if (!lying) rew = a.dish_rew;
    }
    // soup after an ingredient goes in (ref:336-348); an empty pot first gets an empty soup
    uint32_t soup = (t & 0xff) == O_NONE ? kEmptySoup : t;
    soup += (out & F_ADD) ? ((hname == O_ONION) ? 0x100u : 0x10000u) : 0u;
    const int sn = static_cast<int>(((soup >> 8) & 0xff) + ((soup >> 16) & 0xff));
    const bool start = (out & F_START_CHECK) && (soup & 0xff) == O_SOUP && (static_cast<int32_t>(soup) < 0) && sn == 3;
    soup = start ? (soup & 0x00FFFFFFu) : soup;
    const uint32_t nt = (out & F_TILE_GETS_HELD) ? held : (out & F_TILE_CLEARED) ? 0u : (out & (F_ADD | F_START_CHECK)) ? soup : t;
    const uint32_t nh = (out & F_HELD_GETS_TILE) ? t : (out & F_HELD_CLEARED) ? 0u : (out & F_HELD_FRESH) ? (kFreshTick | ((out >> 12) & 7)) : held;
    rew += static_cast<int>(out >> 16) + ((out & F_SERVE) ? rv : 0);
    held = nh;
    t = nt;
    return rew;
}

__device__ __forceinline__ int recipe_index(uint32_t obj) { return (4 * ((obj >> 8) & 0xff) + ((obj >> 16) & 0xff)) & 15; }

template <bool ACT64>
__device__ __forceinline__ int load_action(const void *actions, int64_t idx) {
    if (ACT64) return static_cast<int>(reinterpret_cast<const long long *>(actions)[idx]);
    return reinterpret_cast<const int *>(actions)[idx];
}
__device__ __forceinline__ int sanitize_action(int a) {
    return (static_cast<unsigned>(a) > 5u) ? A_STAY : a;   // out-of-range codes behave as STAY
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// one global_store_dwordx4: the caller guarantees 16-byte alignment (VEC16 launches only)
__device__ __forceinline__ void store16(int8_t *dst, uint4 v) {
    const u32x4 t = {v.x, v.y, v.z, v.w};
    *reinterpret_cast<u32x4 *>(__builtin_assume_aligned(dst, 16)) = t;
}
__device__ __forceinline__ void store4(int8_t *dst, uint32_t v) { *reinterpret_cast<uint32_t *>(dst) = v; }

// Workgroup == one wavefront, and a wavefront's LDS operations execute in issue order, so lanes
// exchange data through LDS with nothing but a compiler-level ordering point.  __syncthreads() would
// also drain vmcnt -- i.e. stall until every plane store and prefetch load in flight has completed.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// held-object contribution to dword 4 (channels 17..19) of the holder's tile (ref:236-253)
__device__ __forceinline__ uint32_t held_flags(uint32_t held) {
    // ONION(2) -> byte 3, DISH(3) -> byte 2, SOUP(4) -> byte 1, NONE/TOMATO -> nothing: low dword of 1 << 8*(5-name)
    return static_cast<uint32_t>(1ull << (8 * ((5 - (held & 7)) & 7)));
}

// One launch = one env step for every kitchen (STEP) or a pure featurisation of the given
// state block (!STEP).  VEC16: every viewer plane base is 16-byte aligned.
template <bool STEP, bool ACT64, bool VEC16, int QN>
__global__ __launch_bounds__(kWave) void oc_step_kernel(const StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const LdsView L = carve(lds, a);
    const int lane = threadIdx.x;
    const int S = a.S;

    // ---- software pipeline: registers hold the *next* group's state block and actions
    // Loads are unconditional (indices clamped into the arrays) so that they land directly in the
    // pipeline registers: a predicated load would be copied -- and therefore waited for -- on the spot.
    uint4 pre[QN];
    int pa0, pa1;
    const int64_t last_chunk = a.n_kitchens * a.recq - 1;
    auto prefetch = [&](int g) {
        const int64_t k0 = static_cast<int64_t>(g) * a.kpw;
        const int64_t c0 = k0 * a.recq + lane;
#pragma unroll
        for (int q = 0; q < QN; ++q) pre[q] = a.state[min(c0 + q * kWave, last_chunk)];
        if (STEP) {
            const int64_t kid = min(k0 + lane, a.n_kitchens - 1);
            pa0 = load_action<ACT64>(a.actions, kid);
            pa1 = load_action<ACT64>(a.actions, a.n_kitchens + kid);
        } else {
            pa0 = pa1 = A_STAY;
        }
    };
    // Make the compiler's vmcnt bookkeeping see the prefetch registers as landed: vmcnt retires in
    // order, so a wait for these loads placed after plane stores have been issued also waits for the stores.
    auto land_prefetch = [&]() {
#pragma unroll
        for (int q = 0; q < QN; ++q) asm volatile("" : "+v"(pre[q].x), "+v"(pre[q].y), "+v"(pre[q].z), "+v"(pre[q].w));
        asm volatile("" : "+v"(pa0), "+v"(pa1));
    };
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    if (a.trace) stamp[0] = __builtin_amdgcn_s_memrealtime();
    // first group's loads go out before anything else so their latency overlaps the table setup
    int group = blockIdx.x;
    prefetch(min(group, a.n_groups - 1));

    // ---- tables -> LDS, once per wave
    for (int i = lane; i < S; i += kWave) {
        L.terrain[i] = a.tabs->terrain[i];
        L.out_tab2[i] = make_uint2(a.tabs->out_tab2[2 * i], a.tabs->out_tab2[2 * i + 1]);
    }
    if (lane < 16) { L.rvalues[lane] = a.tabs->recipe_values[lane]; L.rtimes[lane] = a.tabs->recipe_times[lane]; }
    if (STEP) {
        for (int i = lane; i < kLutSize; i += kWave) L.ilut[i] = a.tabs->interact_lut[i];
        if (lane < a.n_pots) L.pots[lane] = a.tabs->pots[lane];
        for (int i = lane; i < a.n_counters; i += kWave) L.counters[i] = a.tabs->counters[i];
    }

    // where this lane's q-th 16-byte chunk of a group's state block lives in LDS (group independent)
    int loff[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int c = lane + q * kWave;
        const int k = c / a.recq;
        loff[q] = k * a.lstride + 4 * (c - k * a.recq);
    }
    // tile walk constants: pass p covers flat tiles p*64 + lane, flat = kitchen*S + output tile
    const int k_first = lane / S, o_first = lane - k_first * S;
    const int dk = kWave / S, dq = kWave - dk * S;

    land_prefetch();
    if (a.trace) stamp[1] = __builtin_amdgcn_s_memrealtime();

    for (; group < a.n_groups; group += gridDim.x) {
        const int64_t k0 = static_cast<int64_t>(group) * a.kpw;
        const int n_valid = static_cast<int>(min(static_cast<int64_t>(a.kpw), a.n_kitchens - k0));
        const int nq = n_valid * a.recq;

        // ---- prefetched state block -> LDS, then start fetching the next group
#pragma unroll
        for (int q = 0; q < QN; ++q)
            if (lane + q * kWave < nq) {
                uint32_t *d = L.state + loff[q];
                d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
            }
        const int a0 = sanitize_action(pa0), a1 = sanitize_action(pa1);
        prefetch(min(group + static_cast<int>(gridDim.x), a.n_groups - 1));     // last iteration: harmless re-read
        wave_sync();

        // ---- phase T: lane = kitchen.  ref:381-385 get_state_transition + wrappers' done/reset rule
        if (lane < n_valid) {
            uint32_t *rec = L.state + lane * a.lstride;
            uint32_t posor = rec[S], h0 = rec[S + 1], h1 = rec[S + 2];
            uint32_t is_done = 0;
            if (STEP && !(a.ablate & 1)) {
                const int64_t kid = k0 + lane;
                int ts = static_cast<int>(rec[S + 3]);
                int pos0 = posor & 0xff, or0 = (posor >> 8) & 0xff, pos1 = (posor >> 16) & 0xff, or1 = posor >> 24;

                // round 1 of LDS reads happened above (posor, hands, timestep); round 2: everything whose
                // address depends only on positions/orientations/actions, issued together
                const int ipos0 = pos0 + dir_delta(or0, a.delta_pack), ipos1 = pos1 + dir_delta(or1, a.delta_pack);
                const int cand0 = pos0 + dir_delta(a0, a.delta_pack), cand1 = pos1 + dir_delta(a1, a.delta_pack);   // STAY/INTERACT: delta 0
                const uint32_t terr_i0 = L.terrain[ipos0], terr_i1 = L.terrain[ipos1];
                const uint32_t terr_c0 = L.terrain[cand0], terr_c1 = L.terrain[cand1];
                uint32_t tile0 = rec[ipos0], tile1 = rec[ipos1];
                const int rv0 = L.rvalues[recipe_index(h0)], rv1 = L.rvalues[recipe_index(h1)];
                // ref:272-281 get_pot_states, taken once before anybody acts (ref:302)
                int pot_states = 0;
                for (int i = 0; i < a.n_pots; ++i) {
                    const uint32_t o = rec[i < 4 ? a.pots4[i] : L.pots[i]];
                    const int tick = static_cast<int32_t>(o) >> 24;
                    const int n = static_cast<int>(((o >> 8) & 0xff) + ((o >> 16) & 0xff));
                    pot_states += ((o & 0xff) != O_NONE && (tick >= 0 || n < 3)) ? 1 : 0;
                }
                // interacts in player order on pre-move positions; player 1 sees player 0's effects
                int rew = interact(a0 == A_INTERACT, terr_i0, tile0, h0, h1, rv0, pot_states, rec, L, a);
                rec[ipos0] = tile0;
                if (ipos1 == ipos0) tile1 = tile0;
                rew += interact(a1 == A_INTERACT, terr_i1, tile1, h1, h0, rv1, pot_states, rec, L, a);
                rec[ipos1] = tile1;

                // ref:393-399 _move_if_direction, ref:356-366 _handle_collisions
                const int no0 = (a0 >= A_STAY) ? or0 : a0, no1 = (a1 >= A_STAY) ? or1 : a1;
                const int np0 = (terr_c0 != T_AIR) ? pos0 : cand0;        // cand == pos (own, walkable tile) when not moving
                const int np1 = (terr_c1 != T_AIR) ? pos1 : cand1;
                const bool blocked = (np0 == np1) || (np0 == pos1 && pos0 == np1);
                if (!blocked) { pos0 = np0; pos1 = np1; }
                or0 = no0; or1 = no1;

                // ref:373-375 timestep; envs/overcooked2_env.py:334 done; vectorenv.py:369-370 reset rule
                ts += 1;
                is_done = ts >= a.horizon ? 1u : 0u;
                posor = pos0 | (or0 << 8) | (pos1 << 16) | (static_cast<uint32_t>(or1) << 24);
                if (is_done) { posor = a.start_posor; h0 = 0; h1 = 0; ts = 0; }
                rec[S] = posor; rec[S + 1] = h0; rec[S + 2] = h1; rec[S + 3] = static_cast<uint32_t>(ts);
                a.reward[kid] = rew;                     // envs/overcooked2_env.py:336-339: sum, to every player
                a.reward[a.n_kitchens + kid] = rew;
                a.done[kid] = static_cast<int32_t>(is_done);
            }
            // per-kitchen values every tile lane of phase E needs, computed once here
            uint32_t *kd = L.kd + lane * kKd;
            kd[0] = (posor & 0xff) | (((posor >> 16) & 0xff) << 8) | (is_done << 16);
            kd[1] = 1u << (8 * ((posor >> 8) & 3));      // ref:221-234 orientation one-hot bytes, player 0
            kd[2] = 1u << (8 * ((posor >> 24) & 3));     //                                       player 1
            kd[3] = held_flags(h0);
            kd[4] = held_flags(h1);
        }
        wave_sync();
        if (a.trace && stamp[2] == 0) stamp[2] = __builtin_amdgcn_s_memrealtime();

        // ---- phase E: lane = tile, in output order o = x*H + y per kitchen (envs/overcooked2_env.py:324)
        const int ntiles = n_valid * S;
        const int npass = (ntiles + kWave - 1) / kWave;
        int k = k_first, o = o_first;
        const int64_t plane_base = k0 * S * kChannels;   // byte offset of this group's kitchens in a viewer plane
        // One tile of phase E is split in two so that two tiles per lane can be in flight: all LDS
        // reads of both tiles are issued before either tile's arithmetic (the kernel runs at 1-2 waves
        // per SIMD, so instruction-level parallelism inside the wave is what hides LDS latency).
        struct TileIn { uint2 tab; uint4 kq; uint32_t hf1; uint32_t obj; uint32_t *cell; };
        uint32_t *const dump = L.kd + 5;                 // spare dword of kitchen 0's derived block
        auto tile_read = [&](int kk, int oo, bool active) {
            TileIn r;
            kk = min(kk, a.kpw - 1);                     // idle lanes of a ragged last pass read in-range LDS
            r.tab = L.out_tab2[oo];
            r.kq = *reinterpret_cast<const uint4 *>(L.kd + kk * kKd);       // pp, oh0, oh1, hf0
            r.hf1 = L.kd[kk * kKd + 4];
            uint32_t *cell = L.state + kk * a.lstride + (r.tab.x & 0xff);
            r.obj = *cell;
            r.cell = active ? cell : dump;               // idle lanes store to a dump word instead of branching
            return r;
        };
        auto tile_encode = [&](const TileIn &r, int slot) {
            const uint32_t t = r.tab.x & 0xff;
            uint32_t obj = r.obj;
            if (STEP) {
                // ref:376-379: every SOUP lying on a tile that is cooking ticks (held ones do not)
                const int tick = static_cast<int32_t>(obj) >> 24;
                if ((obj & 0xff) == O_SOUP && tick >= 0 && tick < cook_time(L, a, obj)) obj += 0x01000000u;
                if (r.kq.x & 0x10000u) obj = 0;          // kitchen was reset this step
                *r.cell = obj;
            }
            const bool here0 = (r.kq.x & 0xff) == t, here1 = ((r.kq.x >> 8) & 0xff) == t;
            const uint32_t A = here0 ? r.kq.y : 0u, B = here1 ? r.kq.z : 0u;      // ref:221-234 orientation one-hots
            const uint32_t e0 = here0 ? 1u : 0u, e1 = here1 ? 1u : 0u;
            const uint32_t hands = (here0 ? r.kq.w : 0u) | (here1 ? r.hf1 : 0u);  // ref:236-253 objects in hands
            const uint32_t name = obj & 0xff;
            const bool inpot = (name == O_SOUP) && (r.tab.x & 0x100u);
            const int tk = static_cast<int32_t>(obj) >> 24;
            // ref:182-211 objects lying on tiles
            const uint32_t lying = inpot ? 0u : held_flags(obj);
            const uint32_t d3 = r.tab.y | (inpot ? ((obj << 16) & 0xFF000000u) : 0u);      // ch15 = onions in the pot
            const uint32_t d4 = (inpot ? static_cast<uint32_t>(max(tk, 0)) : 0u) | lying | hands;
            const uint32_t thi = r.tab.x >> 16;          // terrain bits of channels 10,11 (ref:165-171)
            uint32_t *s0 = L.stage[0] + slot, *s1 = L.stage[1] + slot;
            // bytes: [self here, other here, self orient x4, other orient x4, terrain ...]
            s0[0] = e0 | (e1 << 8) | (A << 16); s0[1] = __builtin_amdgcn_alignbit(B, A, 16); s0[2] = __builtin_amdgcn_alignbit(thi, B, 16); s0[3] = d3; s0[4] = d4;
            s1[0] = e1 | (e0 << 8) | (B << 16); s1[1] = __builtin_amdgcn_alignbit(A, B, 16); s1[2] = __builtin_amdgcn_alignbit(thi, A, 16); s1[3] = d3; s1[4] = d4;
        };
        auto advance = [&]() {
            k += dk; o += dq;
            if (o >= S) { o -= S; ++k; }
        };
        auto encode_passes = [&](int p0) {
            const int pend = min(p0 + a.gpasses, npass);
            if (a.ablate & 4) return;
            int p = p0;
            for (; p + 1 < pend; p += 2) {               // two tiles per lane in flight
                const int slot = ((p - p0) * kWave + lane) * kTileDwords;
                const TileIn r0 = tile_read(k, o, p * kWave + lane < ntiles);
                advance();
                const TileIn r1 = tile_read(k, o, (p + 1) * kWave + lane < ntiles);
                advance();
                tile_encode(r0, slot);
                tile_encode(r1, slot + kWave * kTileDwords);
            }
            if (p < pend) {
                const TileIn r0 = tile_read(k, o, p * kWave + lane < ntiles);
                advance();
                tile_encode(r0, ((p - p0) * kWave + lane) * kTileDwords);
            }
        };
        auto flush_passes = [&](int p0) {
            // ---- flush the staged tiles of both viewers: contiguous bytes in the output planes
            const int gbytes = (min(ntiles - p0 * kWave, a.gpasses * kWave)) * kChannels;
            const int64_t goff = plane_base + static_cast<int64_t>(p0) * kWave * kChannels;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (a.ablate & 2) break;
                int8_t *dst = a.obs + v * a.obs_player_stride + goff;
                const uint32_t *src = L.stage[v];
                if (VEC16) {
                    const int nfull = gbytes >> 4;            // whole 16-byte chunks: 1 KiB per wave store
                    for (int c = lane; c < nfull; c += kWave) store16(dst + c * 16, *reinterpret_cast<const uint4 *>(src + 4 * c));
                    const int rem = (gbytes & 15) >> 2;       // 0..3 dwords, ragged tail of the very last group only
                    if (lane < rem) store4(dst + nfull * 16 + lane * 4, src[nfull * 4 + lane]);
                } else {
                    for (int c = lane; c * 4 < gbytes; c += kWave) store4(dst + c * 4, src[c]);
                }
            }
        };
        // first flush group peeled so that the prefetch lands at a fixed program point: after the first
        // tiles are encoded (latency cover) and before this group's first plane store is issued
        encode_passes(0);
        wave_sync();
        land_prefetch();
        flush_passes(0);
        wave_sync();
        if (a.trace && stamp[3] == 0) stamp[3] = __builtin_amdgcn_s_memrealtime();
        for (int p0 = a.gpasses; p0 < npass; p0 += a.gpasses) {
            encode_passes(p0);
            wave_sync();
            flush_passes(p0);
            wave_sync();
        }

        if (a.trace && stamp[4] == 0) stamp[4] = __builtin_amdgcn_s_memrealtime();
        // ---- state block LDS -> HBM
        if (STEP && !(a.ablate & 8)) {
            uint4 *dst = a.state + k0 * a.recq;
#pragma unroll
            for (int q = 0; q < QN; ++q)
                if (lane + q * kWave < nq) {
                    const uint32_t *sp = L.state + loff[q];
                    dst[lane + q * kWave] = make_uint4(sp[0], sp[1], sp[2], sp[3]);
                }
        }
        wave_sync();                                 // next iteration overwrites the LDS state block
    }
    if (a.trace) {
        stamp[5] = __builtin_amdgcn_s_memrealtime();
        if (lane < 6) a.trace[static_cast<size_t>(blockIdx.x) * 8 + lane] = stamp[lane];
        if (lane == 6) a.trace[static_cast<size_t>(blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID
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
    int pots4[4];
    uint32_t start_posor;
    int kpw, gpasses, waves_per_cu, num_cus, ablate;
    uint4 *d_state;
    DevTables *d_tabs;
    unsigned long long *d_trace;
    hipEvent_t ev0, ev1;
};

namespace {

size_t lds_bytes(const oc_engine *e, int kpw, int gpasses) {
    const int lstride = 4 * e->recq + 1;
    return sizeof(uint32_t) * (static_cast<size_t>(lds_table_dwords(e->S, e->n_pots, e->n_counters)) + kpw * kKd +
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
    a.n_groups = static_cast<int32_t>((n + e->kpw - 1) / e->kpw);
    a.horizon = e->params.horizon;
    a.placement_rew = e->params.placement_in_pot_rew;
    a.dish_rew = e->params.dish_pickup_rew;
    a.soup_rew = e->params.soup_pickup_rew;
    a.n_pots = e->n_pots;
    a.n_counters = e->n_counters;
    for (int i = 0; i < 4; ++i) a.pots4[i] = e->pots4[i];
    {
        const unsigned long long w = static_cast<unsigned char>(e->params.width), mw = static_cast<unsigned char>(-e->params.width);
        a.delta_pack = mw | (w << 8) | (1ull << 16) | (0xFFull << 24);
    }
    a.uniform_time = e->uniform_time;
    a.start_posor = e->start_posor;
    a.ablate = e->ablate;
    a.trace = e->d_trace;
    return a;
}

template <bool STEP, bool ACT64, bool VEC16, int QN>
void launch_q(const StepArgs &a, unsigned grid, size_t shmem, hipStream_t s) {
    hipLaunchKernelGGL((oc_step_kernel<STEP, ACT64, VEC16, QN>), dim3(grid), dim3(kWave), shmem, s, a);
}

template <bool STEP, bool ACT64, bool VEC16>
void launch_v(const StepArgs &a, unsigned grid, size_t shmem, hipStream_t s) {
    switch ((a.kpw * a.recq + kWave - 1) / kWave) {     // 16-byte chunks per lane per group
    case 1: case 2: launch_q<STEP, ACT64, VEC16, 2>(a, grid, shmem, s); break;
    case 3: launch_q<STEP, ACT64, VEC16, 3>(a, grid, shmem, s); break;
    case 4: launch_q<STEP, ACT64, VEC16, 4>(a, grid, shmem, s); break;
    default: launch_q<STEP, ACT64, VEC16, kMaxQ>(a, grid, shmem, s); break;
    }
}

template <bool STEP, bool ACT64>
int launch(const oc_engine *e, const StepArgs &a, hipStream_t s) {
    if (a.n_kitchens == 0) return OC_OK;
    const bool vec16 = (reinterpret_cast<uintptr_t>(a.obs) % 16 == 0) && (a.obs_player_stride % 16 == 0);
    // persistent waves: at most waves_per_cu per CU, each walking its groups in a grid-stride loop
    const unsigned grid = static_cast<unsigned>(min(static_cast<int64_t>(a.n_groups), static_cast<int64_t>(e->waves_per_cu) * e->num_cus));
    const size_t shmem = lds_bytes(e, a.kpw, a.gpasses);
    if (vec16) launch_v<STEP, ACT64, true>(a, grid, shmem, s);
    else launch_v<STEP, ACT64, false>(a, grid, shmem, s);
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
    if (p->soup_pickup_rew < 0 || p->soup_pickup_rew > 32767 || p->placement_in_pot_rew < 0 || p->placement_in_pot_rew > 32767)
        return fail(OC_ERR_UNSUPPORTED, "oc_create: shaped rewards must be in 0..32767");

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
        {
            const int o = x * p->height + y;
            const uint32_t hi2 = terr == T_POT ? 0x0001u : terr == T_COUNTER ? 0x0100u : 0u;      // channels 10, 11
            tabs.out_tab2[2 * o] = static_cast<uint32_t>(t) | (terr == T_POT ? 0x100u : 0u) | (hi2 << 16);
            tabs.out_tab2[2 * o + 1] = terr >= T_ONION_SRC ? (1u << (8 * (terr - T_ONION_SRC))) : 0u;   // channels 12..15
        }
        if (terr == T_POT) {
            if (n_pots == kMaxPots) return fail(OC_ERR_UNSUPPORTED, "oc_create: more than %d pots", kMaxPots);
            tabs.pots[n_pots++] = t;
        }
        if (terr == T_COUNTER) tabs.counters[n_counters++] = t;
    }
    for (uint32_t terr = 0; terr < 7; ++terr)
        for (uint32_t h = 0; h < 5; ++h)
            for (int c = 0; c < kTileClasses; ++c) tabs.interact_lut[(terr * 5 + h) * kTileClasses + c] = interact_outcome(terr, h, c, p->soup_pickup_rew, p->placement_in_pot_rew);
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
    int num_cus = 0;
    OC_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, device_id));

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
    for (int i = 0; i < 4; ++i) e->pots4[i] = i < n_pots ? static_cast<int>(tabs.pots[i]) : 0;
    e->start_posor = start_posor;
    e->kpw = 16;
    while (e->kpw > 4 && e->kpw * e->recq > kMaxQ * kWave) e->kpw -= 4;
    e->gpasses = 2;
    e->waves_per_cu = 8;
    e->num_cus = num_cus > 0 ? num_cus : 256;
    { const char *ab = getenv("OC_ABLATE"); e->ablate = ab ? atoi(ab) : 0; }
    e->d_state = nullptr;
    e->d_tabs = nullptr;
    e->d_trace = nullptr;
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
    if (e->d_trace) (void)hipFree(e->d_trace);
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

int oc_set_tuning(oc_engine *e, int kitchens_per_wave, int passes_per_flush, int waves_per_cu) {
    if (int rc = check_engine(e)) return rc;
    const int kpw = kitchens_per_wave ? kitchens_per_wave : e->kpw;
    const int gp = passes_per_flush ? passes_per_flush : e->gpasses;
    const int wpc = waves_per_cu ? waves_per_cu : e->waves_per_cu;
    if (kpw < 4 || kpw > 64 || kpw % 4) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: kitchens_per_wave must be a multiple of 4 in 4..64");
    if (kpw * e->recq > kMaxQ * kWave)
        return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: kitchens_per_wave %d x %d B records exceeds the %d B prefetch window", kpw, e->recq * 16, kMaxQ * kWave * 16);
    if (gp < 1 || gp > 8) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: passes_per_flush must be in 1..8");
    if (wpc < 1 || wpc > 32) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: waves_per_cu must be in 1..32");
    if (lds_bytes(e, kpw, gp) > 64 * 1024) return fail(OC_ERR_INVALID_ARG, "oc_set_tuning: %zu B of LDS per wave exceeds 64 KiB", lds_bytes(e, kpw, gp));
    e->kpw = kpw;
    e->gpasses = gp;
    e->waves_per_cu = wpc;
    return OC_OK;
}

/* Diagnostics, deliberately not in the public header: per-wave phase timestamps of the next launches. */
int oc_debug_trace(oc_engine *e, int enable, unsigned long long *host_out, int64_t max_waves) {
    if (int rc = check_engine(e)) return rc;
    const int64_t cap = 32 * 1024;
    if (enable) {
        if (!e->d_trace) OC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_trace), cap * 8 * sizeof(unsigned long long)));
        OC_HIP(hipMemset(e->d_trace, 0, cap * 8 * sizeof(unsigned long long)));
        return OC_OK;
    }
    if (host_out && e->d_trace) {
        OC_HIP(hipDeviceSynchronize());
        OC_HIP(hipMemcpy(host_out, e->d_trace, static_cast<size_t>(max_waves < cap ? max_waves : cap) * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (e->d_trace) { (void)hipFree(e->d_trace); e->d_trace = nullptr; }
    return OC_OK;
}

}  // extern "C"
