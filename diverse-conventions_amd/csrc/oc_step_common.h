// oc_step_common.h -- layout-independent part of the fused Overcooked step + encode kernel (gfx950): codes,
// the INTERACT outcome table, the argument block, small device helpers.  Included once per translation unit;
// oc_step_body.h (the kernel body proper) is included once per layout specialisation after it.
//
// The device code is compiled ahead of time by hipcc as part of overcooked_engine.hip (generic: every layout
// parameter is a runtime field of StepArgs) and again by hiprtc with an engine's parameters as constants
// (OC_JIT; the text of both headers is embedded in the library by the build) -- one specialisation per
// engine (oc_create) or several in one module for a multi-layout launch (oc_group_create).
// Semantics follow the reference's Python MDP (envs/overcooked2_reimplement.py, DummyMDP; "ref:" line
// numbers are of that file unless another file is named).  Must stay free of host-only includes.
#ifndef OC_STEP_COMMON_H
#define OC_STEP_COMMON_H

#ifdef __HIPCC_RTC__
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace oc_dev {

// ---- parameter access: runtime fields of StepArgs in the ahead-of-time build, literal constants
// (-DOC_S=20 ...) when the kernel is compiled for one engine by hiprtc (OC_JIT).  With constants the
// compiler folds all index arithmetic, unrolls the small loops and needs no SGPRs for them: the generic
// build spends ~1 us per wave on scalar set-up and SGPR spills that the specialised build does not have.
#ifdef OC_JIT
#define KP_S(a) (OC_S)
#define KP_recq(a) (OC_RECQ)
#define KP_lstride(a) (4 * OC_RECQ + 1)
#define KP_kpw(a) (OC_KPW)
#define KP_gpasses(a) (OC_GP)
#define KP_kpp(a) (OC_KPP)
#define KP_n_groups(a) ((a).n_groups)
#define KP_horizon(a) (OC_HORIZON)
#define KP_dish_rew(a) (OC_DISH_REW)
#define KP_n_pots(a) (OC_NPOTS)
#define KP_n_counters(a) (OC_NCOUNTERS)
#define KP_uniform_time(a) (OC_UNIFORM_TIME)
#define KP_start_posor(a) (OC_START_POSOR)
#define KP_delta_pack(a) (OC_DELTA_PACK)
#define KP_n_kitchens(a) ((a).n_kitchens)
#if defined(OC_JIT_ABLATE)
#define KP_ablate(a) (OC_JIT_ABLATE)       // diagnostics build: the phase-skipping bits at the time the module is built
#else
#define KP_ablate(a) (0)
#endif
#if defined(OC_TRACE) && OC_TRACE
#define KP_trace(a) ((a).trace)
#else
#define KP_trace(a) (static_cast<unsigned long long *>(nullptr))
#endif
#define KP_pots4(a, i) ((i) == 0 ? OC_POT0 : (i) == 1 ? OC_POT1 : (i) == 2 ? OC_POT2 : OC_POT3)
#else
#define KP_S(a) ((a).S)
#define KP_recq(a) ((a).recq)
#define KP_lstride(a) ((a).lstride)
#define KP_kpw(a) ((a).kpw)
#define KP_gpasses(a) ((a).gpasses)
#define KP_kpp(a) ((a).kpp)
#define KP_n_groups(a) ((a).n_groups)
#define KP_horizon(a) ((a).horizon)
#define KP_dish_rew(a) ((a).dish_rew)
#define KP_n_pots(a) ((a).n_pots)
#define KP_n_counters(a) ((a).n_counters)
#define KP_uniform_time(a) ((a).uniform_time)
#define KP_start_posor(a) ((a).start_posor)
#define KP_delta_pack(a) ((a).delta_pack)
#define KP_n_kitchens(a) ((a).n_kitchens)
// phase ablation and the per-wave phase stamps exist in the diagnostics build only (-DOC_DIAGNOSTICS: csrc/liboc_hip_diag.so, tools/):
// the product library's kernels have neither the branches nor a way to switch them on
#ifdef OC_DIAGNOSTICS
#define KP_ablate(a) ((a).ablate)
#define KP_trace(a) ((a).trace)
#else
#define KP_ablate(a) (0)
#define KP_trace(a) (static_cast<unsigned long long *>(nullptr))
#endif
#define KP_pots4(a, i) ((a).pots4[i])
#endif

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

// Per-layout lookup tables: one flat image in HBM (L2-hot), laid out exactly like the table part of
// the LDS carve (see carve()) and copied to LDS by every wave:
//   out_tab2[2S]  by output tile index o = x*H + y, two dwords:
//                 [0] t | is_pot<<8 | (terrain bits of channels 10,11 as the high half of dword 2)<<16
//                 [1] terrain bits of channels 12..15 (dword 3)
//   terrain[S]    by state tile index t = y*W + x
//   recipe_values[16] (ref:283-285), recipe_times[16] (ref:156-157)
//   interact_lut[kLutSize]  outcome of INTERACT by (terrain, held name, faced-tile class)
//   pots[n_pots], counters[n_counters]  tile indices with terrain POT / COUNTER
typedef uint32_t DevTables;

struct StepArgs {
    uint4 *state;               // [N][recq] uint4
    const DevTables *tabs;      // table image, lds_table_dwords() dwords
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
    int32_t gpasses;            // passes staged per flush
    int32_t kpp;                // > 0: kitchen-aligned passes, kpp whole kitchens (kpp*S <= 64 tiles) per pass; 0: flat 64-tile passes
    int32_t n_groups;           // ceil(n_kitchens / kpw)
    int32_t horizon, placement_rew, dish_rew, soup_rew;
    int32_t n_pots, n_counters;
    int32_t pots4[4];           // first four POT tile indices (rest: DevTables::pots)
    unsigned long long delta_pack;  // signed bytes {-W, +W, +1, -1, 0, 0}: tile-index delta per direction/action
    int32_t uniform_time;       // recipe time if all 16 are equal, else -1
    uint32_t start_posor;       // p0.pos | NORTH<<8 | p1.pos<<16 | NORTH<<24   (ref:387-391)
    // fused multi-step rollout (FUSED kernels only): every wave keeps its kitchens' state in LDS for n_steps
    // steps; step t reads action slice (t % tape_steps) or draws actions from the counter hash (actions == NULL)
    int32_t n_steps, tape_steps;
    int64_t action_step_stride; // elements between consecutive tape slices (= P*N)
    int64_t obs_step_stride;    // bytes added to obs per step (0 = overwrite in place)
    int64_t reward_step_stride, done_step_stride;   // elements
    uint32_t rng_seed, rng_t0;  // hash RNG: action = hash(seed, kitchen, rng_t0 + t, player) % 6
    // batched featurisation (BATCH kernels only): n_groups = batch_items * batch_groups, see step_body
    int32_t batch_groups;       // groups per batch item = ceil(n_kitchens / kpw)
    int64_t state_batch_stride; // uint4 between consecutive items' record blocks
    int64_t obs_batch_stride;   // bytes between consecutive items' viewer-0 planes
    // trajectory recording (STEP kernels): the post-step state records also go to rec_out (same [N][recq] layout as `state`, 16-byte
    // aligned), step t of a FUSED rollout to rec_out + t * rec_step_stride; NULL = not recorded
    uint4 *rec_out;
    int64_t rec_step_stride;    // uint4 between consecutive steps' record blocks (FUSED kernels only)
    // builds whose groups read a neighbour's kitchen (OC_HALO_R, OC_XW_SUM: oc_step_body.h): where the post-step records go instead
    // of `state` -- a second block of the same size, swapped with the first by the host after the launch; NULL = in place, no halo
    uint4 *state_out;
    int32_t ablate;             // diagnostics build only (tools/ablate.py): phases to skip; always 0 in the product library
    unsigned long long *trace;  // diagnostics build only: [wave][8] s_memrealtime stamps (100 MHz); always NULL in the product library
    // TWOPTR kernels only (oc_step_players, round 6): player 1's actions [N], a buffer of its own -- `actions` is then player 0's [N].
    // Every other kernel reads player 1's at actions + N (one [P][N] tensor) and never looks at this field.
    const void *actions1;
};

// One launch over several engines (oc_group_step / oc_group_observe): blocks [first_block[s], first_block[s+1]) of the
// grid serve segment s with the body compiled for layout block seg_layout[s] of the group module.
constexpr int kMaxSegments = 8;
struct GroupArgs {
    int32_t n_segs;
    int32_t seg_layout[kMaxSegments];
    int32_t first_block[kMaxSegments + 1];
    StepArgs seg[kMaxSegments];
};

constexpr int kMaxQ = 5;        // prefetch registers: up to 5 uint4 per lane => kpw * recq <= 320 (template QN <= kMaxQ)
constexpr int kKd = 8;
constexpr int kTabQ = 2;        // table image chunks per lane covered by the unrolled copy (2 KiB; more -> loop)

__host__ __device__ inline int align4(int x) { return (x + 3) & ~3; }

__host__ __device__ inline int lds_table_dwords(int S, int n_pots, int n_counters) {
    return align4(3 * S + 32 + kLutSize + n_pots + n_counters);
}

// ref:22-32 move_in_direction: tile-index delta of direction/action d in 0..5, looked up in six signed
// bytes {-W, +W, +1, -1, 0, 0} (a select chain here is lowered to a divergent branch tree by the compiler)
__device__ __forceinline__ int dir_delta(int d, unsigned long long pack) {
    return static_cast<int>(static_cast<signed char>(pack >> (8 * d)));
}

__device__ __forceinline__ int recipe_index(uint32_t obj) { return (4 * ((obj >> 8) & 0xff) + ((obj >> 16) & 0xff)) & 15; }

// COHERENT: the actions were written earlier in this very launch by wavefronts of other XCDs (the env step folded into the policy
// launch): an agent-scope load, which does not trust this XCD's L2 or the CU's L1 the way a plain load does between launches.
template <bool ACT64, bool COHERENT = false>
__device__ __forceinline__ int load_action(const void *actions, int64_t idx) {
    if (COHERENT) {
        if (ACT64) return static_cast<int>(__hip_atomic_load(reinterpret_cast<const long long *>(actions) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        return __hip_atomic_load(reinterpret_cast<const int *>(actions) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ACT64) return static_cast<int>(reinterpret_cast<const long long *>(actions)[idx]);
    return reinterpret_cast<const int *>(actions)[idx];
}
// Counter-based action stream for synthetic rollouts: a pure function of (seed, kitchen, step,
// player), so the CPU oracle can be fed the identical stream (tests replicate it in numpy).
__host__ __device__ inline uint32_t oc_hash_action(uint32_t seed, uint32_t kitchen, uint32_t t, uint32_t player) {
    uint32_t x = seed ^ (kitchen * 0x9E3779B1u) ^ (t * 0x85EBCA77u) ^ (player * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return static_cast<uint32_t>((static_cast<unsigned long long>(x) * 6ull) >> 32);
}

__device__ __forceinline__ int sanitize_action(int a) {
    return (static_cast<unsigned>(a) > 5u) ? A_STAY : a;   // out-of-range codes behave as STAY
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// one global_store_dwordx4 (VEC16 kernels).  SM = store mode of the observation planes:
//   0 plain: the lines stay dirty in the XCD's L2 until evicted or until the end-of-kernel write-back;
//   1 non-temporal, for destinations that are not rewritten soon (a fresh trajectory slot every step): measured 13.5
//     vs 14.7 us per launch into a 10 GB time-major buffer, but 13.5 vs 13.2 us when the same 52 MB buffer is
//     overwritten in place (it then lives in the Infinity Cache);
//   2 / 3 write-through (sc1 / sc0 sc1): the bytes leave L2 as they are written instead of waiting, dirty, for the
//     write-back at the kernel boundary -- for batches whose whole output fits the 32 MB of L2 that write-back is
//     otherwise a serial tail of bytes / 6 TB/s behind the last wavefront (MI355X_MICROARCH.md, "boundary" row).
// The write-through forms have no builtin for 16-byte stores, hence the inline asm; nothing in the kernel reads what
// these stores write, so the asm needs no memory clobber, and stores the compiler does not count only make its own
// vmcnt waits more conservative (vmcnt retires in order).  Every asm store is followed by `s_nop 0`: the hazard
// recogniser does not look inside asm, and gfx9-family parts need one wait state between a store of more than 64 bits
// and a VALU write of its data registers (the 12-dword-store hazard) -- the step body is compiled anew per layout and
// tuning, so the register allocation around the flush is not the same in every build.
enum : int { SM_PLAIN = 0, SM_NT = 1, SM_SC1 = 2, SM_SC0SC1 = 3, SM_SC1_NT = 4 };
// The destination is 16-byte aligned relative to its viewer plane, and a plane is only guaranteed 4-byte alignment (the
// second viewer's plane of 8190 kitchens of a 5x5 grid starts 8 bytes off): gfx950 under ROCm runs compute queues in
// unaligned-access mode, where a dwordx4 access needs dword alignment only -- measured 14.98 us per launch against 17.47 us
// for the same batch written with dword stores (profiles/r02_misaligned_planes.log).
typedef u32x4 __attribute__((aligned(4))) u32x4_dword_aligned;
template <int SM>
__device__ __forceinline__ void store16(int8_t *dst, uint4 v) {
    const u32x4 t = {v.x, v.y, v.z, v.w};
    if (SM == SM_NT) __builtin_nontemporal_store(t, reinterpret_cast<u32x4_dword_aligned *>(dst));
    else if (SM == SM_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" : : "v"(dst), "v"(t));
    else if (SM == SM_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" : : "v"(dst), "v"(t));
    else if (SM == SM_SC1_NT) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 0" : : "v"(dst), "v"(t));
    else *reinterpret_cast<u32x4_dword_aligned *>(dst) = t;
}
// The same store addressed as scalar base + 32-bit lane offset + immediate (IMM < 4096): no 64-bit vector address to form and
// carry per store.  `sbase` must be wave-uniform.
template <int SM, int IMM>
__device__ __forceinline__ void store16_s(int8_t *sbase, uint32_t voff, uint4 v) {
    const u32x4 t = {v.x, v.y, v.z, v.w};
    if (SM == SM_NT) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt\n\ts_nop 0" : : "v"(voff), "v"(t), "s"(sbase), "n"(IMM) : "memory");
    else if (SM == SM_SC1) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 0" : : "v"(voff), "v"(t), "s"(sbase), "n"(IMM) : "memory");
    else if (SM == SM_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc0 sc1\n\ts_nop 0" : : "v"(voff), "v"(t), "s"(sbase), "n"(IMM) : "memory");
    else if (SM == SM_SC1_NT) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc1 nt\n\ts_nop 0" : : "v"(voff), "v"(t), "s"(sbase), "n"(IMM) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 0" : : "v"(voff), "v"(t), "s"(sbase), "n"(IMM) : "memory");
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

}  // namespace oc_dev

#endif  // OC_STEP_COMMON_H
