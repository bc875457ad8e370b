// oc_np_kernels.h -- kitchens with 1, 3 or 4 players (gfx950).
//
// The reference's Python MDP and encoder are N-player (envs/overcooked2_reimplement.py: per-viewer player slots :221-234,
// the dish rule that only pays with exactly two players :261-263; envs/overcooked2_env.py:29 `num_players`; layouts
// multiplayer_schelling and simple_single); its training scripts and the north star of this engine are 2-player, and the
// hand-tuned step kernel (oc_step_body.h) is built around two players (packed positions, 20 channels = 5 dwords per tile,
// the viewer swap).  Other player counts run on the group kernel below: the same execution model -- one launch per step, one
// persistent wavefront per group of kitchens, records in LDS, planes flushed as contiguous 16-byte chunks -- with the encoding
// arranged for ANY number of viewers (the tile's viewer-independent bytes written once per viewer, the players' bytes poked in).
// Same semantics, same buffers and layouts with P and C = 5P + 10 generalised; bit-exact against fixtures recorded from the
// reference (tests/golden/np_*.npz) and against the oracle on batches.  Ahead-of-time build, tile count and grid sides at run time.
// "ref:" = envs/overcooked2_reimplement.py.
//
// State record, little endian, 16 * ceil((S + 2P + 1) / 4) bytes:
//   uint32 tile[S]      name | onions<<8 | tomatoes<<16 | (uint8)cooking_tick<<24      (0 = empty)
//   uint32 player[P]    position | orientation<<8
//   uint32 held[P]      packed like a tile object
//   uint32 timestep
#pragma once

#include "oc_step_common.h"

namespace oc_np {

using namespace oc_dev;

constexpr int kMaxP = 4;

struct NpArgs {
    uint32_t *state;                // [N][recd]
    const uint32_t *tabs;           // the engine's table image (prepare_engine): [out_tab2 2S | terrain S | recipe values 16 | recipe times 16 | ...];
                                    // low byte of out_tab2[2 * o] = state tile index t of output tile o = x * H + y
    const void *actions;            // [P][N] i32 or i64 (NULL: encode only)
    int act64;
    int8_t *obs;                    // viewer v at obs + v * obs_player_stride; [N][W][H][C]
    int64_t obs_player_stride;
    int32_t *reward, *done;
    int64_t n;
    int32_t W, H, S, P, C, recd;
    int32_t horizon, placement_rew, soup_rew;
    uint32_t start_pp[kMaxP];
    uint32_t *rec_out;              // the post-step (STEP) / the featurised (!STEP) records also go here ([N][recd], 16-byte aligned: the trajectory's slot); NULL: not recorded
    int32_t K, G;                   // launch geometry: kitchens per group, passes of 64 tiles per flush (NpGeom)
    // batched featurisation (!STEP: oc_encode_trajectory): `items` blocks of n records each, item b at state + b * state_item_stride
    // (dwords) with its planes at obs + b * obs_item_stride (bytes); groups run over all items.  1 item otherwise.
    int64_t items, state_item_stride, obs_item_stride;
    // K steps per launch (STEP: oc_rollout_fused): a group's records stay in LDS for n_steps steps; step t reads action slice
    // t % tape_steps of a tape of [P][N] slices and writes obs + t * obs_step_stride (bytes; 0 = in place), reward + t * reward_step_stride,
    // done + t * done_step_stride (elements), rec_out + t * rec_step_stride (dwords).  n_steps = 1 otherwise.
    int32_t n_steps, tape_steps;
    int64_t obs_step_stride, reward_step_stride, done_step_stride, rec_step_stride;
    int32_t no_pipeline;            // diagnostics: records loaded at the head of each group instead of one group ahead
    float inv_S, inv_H, inv_W, inv_recq;   // 1 / S ...: quotients of the small indices below as (x + 0.5) * inv (np_div; exact there)
};

__device__ __forceinline__ int np_delta(int d, int W) { return d == A_NORTH ? -W : d == A_SOUTH ? W : d == A_EAST ? 1 : d == A_WEST ? -1 : 0; }   // ref:22-32
__device__ __forceinline__ int np_tick(uint32_t o) { return static_cast<int32_t>(o) >> 24; }
__device__ __forceinline__ int np_count(uint32_t o) { return static_cast<int>(((o >> 8) & 0xff) + ((o >> 16) & 0xff)); }

// ---- one launch per step (round 5; rounds 2-4: a thread-per-kitchen transition launch + a block-per-few-kitchens encode launch) ------------
// The 2-player kernel's execution model (oc_step_body.h) with everything player-count specific re-thought rather than generalised:
// one wavefront per workgroup, persistent, a grid-stride loop over GROUPS of K kitchens whose records live in LDS from the
// coalesced load to the coalesced write-back.
//   phase T  lane = kitchen: ref:381-385 get_state_transition on the LDS record (interacts in player order, movement with the
//            collision rule, timestep, done / reset of the players), reward and done to HBM.
//   phase E  lane = tile, 64 tiles per pass, G passes per flush: soups tick / the reset clears the tile (ref:373-379, in the record),
//            then the tile's VIEWER-INDEPENDENT bytes -- ten of its C = 5P + 10: terrain one-hot, pot contents, objects lying on it
//            (ref:165-211) -- behind 5P zero bytes are written to every viewer's staging range: the P planes differ in the player
//            bytes only.  Tiles are staged back to back (tile f of the window at byte f * C -- 15 / 25 / 30 bytes, so most tiles
//            start off a dword: LDS takes dword / 96-bit / 128-bit writes at any byte address on gfx950,
//            tools/probes/lds_unaligned.hip), so that the flush reads each viewer's bytes as aligned 16-byte chunks.
//   pokes    lane = (kitchen, player): the bytes a player contributes -- "here" and orientation in the slot the viewer gives that
//            player (ref:221-234), the object in its hands (ref:236-253) -- are single bytes of value 1 written into the staged
//            tile it stands on, for every viewer, if that tile is in the window: 3 byte stores per viewer and player, no
//            read-modify-write, after the tile writes (a wavefront's LDS operations execute in order).
//   flush    both as the 2-player kernel: contiguous bytes of each viewer's plane, 16 bytes per lane.
// What this replaces cost 108.8 us per step at 4 x 65 536 kitchens of multiplayer_schelling (21.6 us of it the transition launch:
// one wavefront per SIMD walking a chain of dependent global loads) and 30.4 us for 65 536 one-player kitchens.
struct NpGeom { int K, G; };                               // kitchens per group, passes (of 64 tiles) per flush
// x / d for 0 <= x < 2^16 and the small divisors of this kernel (tiles, grid sides, chunks per record <= 255), inv = 1.0f / d:
// (x + 0.5) / d is at least 0.5 / d away from an integer, the f32 product's error stays below 2^-20 of it
__device__ __forceinline__ int np_div(int x, float inv) { return static_cast<int>((static_cast<float>(x) + 0.5f) * inv); }
__host__ __device__ inline int np_lstride(int recd) { return recd | 1; }            // odd: lanes = kitchens hit distinct banks
// LDS of one wavefront, in dwords: [table image: out_tab2 2S | terrain S | rvalues 16 | rtimes 16, whole 16-byte chunks][tab S][state K * lstride]
// [done K][dump 16][staging P * G * 64 * C bytes]
__host__ __device__ inline int np_img_dwords(int S) { return (3 * S + 32 + 3) & ~3; }
__host__ __device__ inline int np_lds_fixed_dwords(int S, int recd, int K) { return (np_img_dwords(S) + S + K * np_lstride(recd) + K + 16 + 3) & ~3; }
__host__ __device__ inline size_t np_lds_bytes(int S, int P, int recd, NpGeom g) {
    return static_cast<size_t>(np_lds_fixed_dwords(S, recd, g.K)) * 4 + static_cast<size_t>(P) * g.G * 64 * (5 * P + 10);
}

template <int P, bool STEP, int SM, bool FUSED = false, bool BATCH = false>
__global__ __launch_bounds__(64) void np_group_kernel(const NpArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t np_lds[];
    constexpr int C = 5 * P + 10, shift = 5 * P;
    const int lane = static_cast<int>(threadIdx.x);
    // Every kernel argument the wavefront will use, asked for at once: left alone the compiler loads each field next to its first use,
    // one scalar-memory round trip after the other in front of the first memory request (five of them in one build: +0.5 us per launch
    // on every shape -- 5.5 -> 6.1 us at 8192 one-player kitchens; seen in the listing, as for the 2-player kernel in round 3).
    asm volatile("" : : "s"(a.state), "s"(a.tabs), "s"(a.actions), "s"(a.obs), "s"(a.reward), "s"(a.done), "s"(a.rec_out), "s"(a.n),
                 "s"(a.obs_player_stride), "s"(a.S), "s"(a.W), "s"(a.H), "s"(a.recd), "s"(a.K), "s"(a.G), "s"(a.act64));
    asm volatile("" : : "s"(a.horizon), "s"(a.placement_rew), "s"(a.soup_rew), "s"(a.inv_S), "s"(a.inv_H), "s"(a.inv_W), "s"(a.inv_recq),
                 "s"(a.no_pipeline), "s"(a.items), "s"(a.state_item_stride), "s"(a.obs_item_stride), "s"(a.n_steps), "s"(a.tape_steps));
    const int S = a.S, W = a.W, H = a.H, recd = a.recd, recq = recd >> 2, K = a.K, G = a.G;
    const int lstride = np_lstride(recd);
    uint32_t *const l_terr = np_lds + 2 * S;               // [S] terrain code of state tile t (the image's terrain section)
    int32_t *const l_rval = reinterpret_cast<int32_t *>(np_lds + 3 * S);
    int32_t *const l_rtime = l_rval + 16;
    uint32_t *const l_tab = np_lds + np_img_dwords(S);     // [S] by output tile o = x * H + y: t | terrain << 8
    uint32_t *const l_state = l_tab + S;                   // [K][lstride]
    uint32_t *const l_done = l_state + K * lstride;        // [K]
    uint32_t *const l_dump = l_done + K;                   // 16 dwords nobody reads
    int8_t *const l_stage = reinterpret_cast<int8_t *>(np_lds + np_lds_fixed_dwords(S, recd, K));    // [P][G * 64 * C]
    const int vstride = G * 64 * C;                        // bytes of one viewer's staging range (a multiple of 64)

    const int64_t groups_per_item = (a.n + K - 1) / K, n_groups = BATCH ? groups_per_item * a.items : groups_per_item;
    // group -> (item, first kitchen inside the item); one item unless a whole trajectory is featurised by one launch (BATCH: a build of
    // its own -- behind a run-time test the compiler still issued the 64-bit division for every group: 12.8 -> 13.5 us per step at
    // 65 536 one-player kitchens)
    auto item_of = [&](int64_t g) { return BATCH ? static_cast<int64_t>(static_cast<uint32_t>(g) / static_cast<uint32_t>(groups_per_item)) : static_cast<int64_t>(0); };
    // ---- software pipeline: registers hold the NEXT group's records and actions while this one is stepped and encoded (the loads
    //      are unconditional, indices clamped into the arrays, so that they land in these registers and are waited for only where
    //      the loop says so -- before a group's first plane store: vmcnt retires in order, a later wait would sit behind the stores)
    constexpr int kNpQ = 4;                                // 16-byte chunks per lane: K * recq <= 256 (np_geometry)
    u32x4 pre[kNpQ];
    int pa[P];
    const bool aligned = (reinterpret_cast<uintptr_t>(a.state) & 15) == 0 && (a.state_item_stride & 3) == 0 && !a.no_pipeline;
    // (clamped to the GROUP's own chunks: lanes past them re-read its last one.  Clamped to the array's end instead they read the
    // following groups' records -- 2.7 x the bytes with sixteen 96-byte records per group, and lines their owners are writing back:
    // 65 536 one-player kitchens 18.3 against 15.3 us per step, tools/ab_nplayer.py)
    auto prefetch = [&](int64_t g) {
        const int64_t item = item_of(g), gi = g - item * groups_per_item;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(BATCH ? a.state + item * a.state_item_stride : a.state) + gi * K * recq;
        const int nkg = static_cast<int>(min(static_cast<int64_t>(K), a.n - gi * K));
        const int room = nkg * recq - 1;
#pragma unroll
        for (int q = 0; q < kNpQ; ++q) pre[q] = src[min(lane + q * 64, room)];
        if (STEP) {
            const int64_t k = gi * K + min(lane, nkg - 1);       // (STEP launches have one item)
#pragma unroll
            for (int i = 0; i < P; ++i)
                pa[i] = a.act64 ? static_cast<int>(reinterpret_cast<const long long *>(a.actions)[i * a.n + k])
                                : reinterpret_cast<const int *>(a.actions)[i * a.n + k];
        }
    };
    auto land_prefetch = [&]() {
#pragma unroll
        for (int q = 0; q < kNpQ; ++q) asm volatile("" : "+v"(pre[q]));
#pragma unroll
        for (int i = 0; i < P; ++i) asm volatile("" : "+v"(pa[i]));
    };
#pragma unroll
    for (int i = 0; i < P; ++i) pa[i] = A_STAY;
    // ---- tables, once per wavefront: the head of the engine's table image as ONE flat 16-byte copy -- its loads go out first (L2
    //      hits), the first group's records and actions (HBM) behind them, and land through an empty asm (as three loops over
    //      terrain / tile map / recipe tables the compiler made three or four serial round trips of it, the first behind the records:
    //      the 2-player kernel's lesson of round 3)
    {
        const u32x4 *img = reinterpret_cast<const u32x4 *>(a.tabs);
        const int nq4 = np_img_dwords(S) >> 2;
        u32x4 tq[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) tq[q] = img[min(lane + q * 64, nq4 - 1)];
        if (aligned && static_cast<int64_t>(blockIdx.x) < n_groups) prefetch(blockIdx.x);     // the first group's loads: in flight during the table setup
#pragma unroll
        for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(tq[q]));
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (lane + q * 64 < nq4) reinterpret_cast<u32x4 *>(np_lds)[lane + q * 64] = tq[q];
        for (int c = lane + 128; c < nq4; c += 64) reinterpret_cast<u32x4 *>(np_lds)[c] = img[c];       // grids of more than 160 tiles
    }
    wave_sync();
    for (int o = lane; o < S; o += 64) {
        const uint32_t t = np_lds[2 * o] & 0xff;
        l_tab[o] = t | (l_terr[t] << 8);
    }
    land_prefetch();
    wave_sync();

    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t item = item_of(g);
        const int64_t k0 = (g - item * groups_per_item) * K;
        const int nk = static_cast<int>(min(static_cast<int64_t>(K), a.n - k0));
        const uint32_t *const item_state = BATCH ? a.state + item * a.state_item_stride : a.state;
        int8_t *const item_obs = BATCH ? a.obs + item * a.obs_item_stride : a.obs;
        int raw[P];
        if (aligned) {
            // ---- prefetched records -> LDS, then start fetching the next group
#pragma unroll
            for (int q = 0; q < kNpQ; ++q) {
                const int c = lane + q * 64;
                if (c < nk * recq) {
                    const int k = np_div(c, a.inv_recq), j = c - k * recq;
                    uint32_t *d = l_state + k * lstride + 4 * j;
                    d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
                }
            }
#pragma unroll
            for (int i = 0; i < P; ++i) raw[i] = pa[i];
            if (g + gridDim.x < n_groups) prefetch(g + gridDim.x);       // (wave-uniform)
        } else {                                           // (!STEP on a caller's records that are only dword aligned: no pipeline)
#pragma unroll
            for (int i = 0; i < P; ++i) raw[i] = A_STAY;
            if (STEP && lane < nk) {
#pragma unroll
                for (int i = 0; i < P; ++i)
                    raw[i] = a.act64 ? static_cast<int>(reinterpret_cast<const long long *>(a.actions)[i * a.n + k0 + lane])
                                     : reinterpret_cast<const int *>(a.actions)[i * a.n + k0 + lane];
            }
            const uint32_t *src = item_state + k0 * recd;
            for (int c = lane; c < nk * recd; c += 64) {
                const int k = np_div(c >> 2, a.inv_recq);
                l_state[k * lstride + (c - k * recd)] = src[c];
            }
        }
        wave_sync();

      // K steps per launch: the whole rollout of this group happens here, its records never leave LDS between steps
      // (FUSED: a build of its own -- as a runtime loop of one iteration it cost the one-step launches 5-20 %, 12.8 -> 14.0 us at
      // 65 536 one-player kitchens)
      const int n_steps = FUSED ? a.n_steps : 1;
      for (int step = 0; step < n_steps; ++step) {
        int raw_next[P];
#pragma unroll
        for (int i = 0; i < P; ++i) raw_next[i] = A_STAY;
        if (FUSED && step + 1 < n_steps && lane < nk) {       // the next step's actions: in flight during this step
            const int64_t slice = static_cast<int64_t>((step + 1) % a.tape_steps) * P * a.n;
#pragma unroll
            for (int i = 0; i < P; ++i)
                raw_next[i] = a.act64 ? static_cast<int>(reinterpret_cast<const long long *>(a.actions)[slice + i * a.n + k0 + lane])
                                      : reinterpret_cast<const int *>(a.actions)[slice + i * a.n + k0 + lane];
        }
        int8_t *const step_obs = FUSED ? item_obs + step * a.obs_step_stride : item_obs;
        int32_t *const step_reward = FUSED ? a.reward + step * a.reward_step_stride : a.reward;
        int32_t *const step_done = FUSED ? a.done + step * a.done_step_stride : a.done;

        // ---- phase T: lane = kitchen
        if (lane < nk) {
            uint32_t *rec = l_state + lane * lstride;
            uint32_t is_done = 0;
            if (STEP) {
                const int64_t k = k0 + lane;
                int pos[P], ori[P], act[P];
                uint32_t held[P];
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const uint32_t pp = rec[S + i];
                    pos[i] = pp & 0xff; ori[i] = (pp >> 8) & 0xff;
                    held[i] = rec[S + P + i];
                    act[i] = sanitize_action(raw[i]);
                }
                int ts = static_cast<int>(rec[S + 2 * P]);
                // ref:301-354 resolve_interacts, players in index order.  get_pot_states (ref:272-281) is only consulted by the dish
                // rule, which pays only when there are exactly two players (ref:262-263): here a dish pick-up is never rewarded.
                int reward = 0;
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    if (act[i] != A_INTERACT) continue;
                    const int ipos = pos[i] + np_delta(ori[i], W);
                    const uint32_t terr = l_terr[ipos];
                    uint32_t t = rec[ipos];
                    const uint32_t hname = held[i] & 0xff;
                    const bool holding = hname != O_NONE, occupied = (t & 0xff) != O_NONE;
                    if (terr == T_COUNTER) {                                                     // ref:313-319
                        if (holding && !occupied) { t = held[i]; held[i] = 0; }
                        else if (!holding && occupied) { held[i] = t; t = 0; }
                    } else if (terr == T_ONION_SRC && !holding) held[i] = kOnionObj;             // ref:320-321
                    else if (terr == T_TOMATO_SRC && !holding) held[i] = kTomatoObj;             // ref:322-323
                    else if (terr == T_DISH_SRC && !holding) held[i] = kDishObj;                 // ref:324-327 (never useful here)
                    else if (terr == T_POT && holding) {                                         // ref:331-349
                        const int time = l_rtime[recipe_index(t)];
                        if (hname == O_DISH && occupied && np_tick(t) >= 0 && np_tick(t) >= time) {
                            held[i] = t; t = 0;
                            reward += a.soup_rew;
                        } else if (hname == O_ONION || hname == O_TOMATO) {
                            if (!occupied) t = kEmptySoup;
                            if (!(np_tick(t) >= 0 || np_count(t) == 3)) {
                                t += hname == O_ONION ? 0x100u : 0x10000u;
                                held[i] = 0;
                                reward += a.placement_rew;
                            }
                            // ref:287-296,347-348: an idle soup with exactly three ingredients starts cooking
                            const int time2 = l_rtime[recipe_index(t)];
                            const bool cooking = np_tick(t) >= 0 && np_tick(t) < time2, ready = np_tick(t) >= 0 && np_tick(t) >= time2;
                            if ((t & 0xff) == O_SOUP && !cooking && !ready && np_count(t) > 0 && np_count(t) == 3) t &= 0x00FFFFFFu;
                        }
                    } else if (terr == T_SERVING && hname == O_SOUP) {                           // ref:350-353, 283-285
                        reward += l_rval[recipe_index(held[i])];
                        held[i] = 0;
                    }
                    rec[ipos] = t;
                }
                // ref:368-371 resolve_movement, ref:393-399 _move_if_direction, ref:356-366 _handle_collisions
                int npos[P], nori[P];
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const bool stays = act[i] == A_INTERACT;
                    const int cand = pos[i] + (stays ? 0 : np_delta(act[i], W));
                    nori[i] = (stays || act[i] == A_STAY) ? ori[i] : act[i];
                    npos[i] = l_terr[cand] != T_AIR ? pos[i] : cand;
                }
                bool blocked = false;
#pragma unroll
                for (int x = 0; x < P; ++x)
#pragma unroll
                    for (int y = x + 1; y < P; ++y) blocked = blocked || npos[x] == npos[y] || (npos[x] == pos[y] && pos[x] == npos[y]);
                // ref:373-375 timestep; envs/overcooked2_env.py:334 done; vectorenv.py:369-370: a finished kitchen restarts at once
                ts += 1;
                is_done = ts >= a.horizon ? 1u : 0u;
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const int np_ = blocked ? pos[i] : npos[i];
                    rec[S + i] = is_done ? a.start_pp[i] : (static_cast<uint32_t>(np_) | (static_cast<uint32_t>(nori[i]) << 8));
                    rec[S + P + i] = is_done ? 0u : held[i];
                    step_reward[i * a.n + k] = reward;                                           // sum, to every player (overcooked2_env.py:336-339)
                }
                rec[S + 2 * P] = is_done ? 0u : static_cast<uint32_t>(ts);
                step_done[k] = static_cast<int32_t>(is_done);
            }
            l_done[lane] = is_done;
        }
        wave_sync();

        // ---- what each (kitchen, player) lane will poke into the staged tiles: flat index of the tile it stands on, its orientation,
        //      the flag byte of what it holds (0: nothing that is encoded)
        int pk_flat = -1, pk_ori = 0, pk_flag = 0, pk_i = 0;
        if (lane < nk * P) {
            const int k = lane / P;                       // (P is a constant of the build)
            pk_i = lane - k * P;
            const uint32_t *rec = l_state + k * lstride;
            const uint32_t pp = rec[S + pk_i], hn = rec[S + P + pk_i] & 0xff;
            const int t = static_cast<int>(pp & 0xff), y = np_div(t, a.inv_W), x = t - y * W;
            pk_flat = k * S + x * H + y;
            pk_ori = static_cast<int>((pp >> 8) & 3);
            pk_flag = hn == O_SOUP ? shift + 7 : hn == O_DISH ? shift + 8 : hn == O_ONION ? shift + 9 : 0;   // ref:236-253 (tomatoes are not encoded)
        }

        // ---- phase E + flush, G passes at a time
        const int ntiles = nk * S;
        const int64_t plane0 = k0 * S * static_cast<int64_t>(C);          // byte offset of the group's first tile in a viewer's plane
        for (int f0 = 0; f0 < ntiles; f0 += G * 64) {
            const int wtiles = min(G * 64, ntiles - f0);
            for (int p = 0; p * 64 < wtiles; ++p) {
                const int fw = p * 64 + lane, f = f0 + fw;            // tile of the window / of the group
                const bool active = fw < wtiles;
                const int fc = active ? f : f0;                       // (idle lanes of a ragged pass read the window's first tile)
                const int k = np_div(fc, a.inv_S), o = fc - k * S;
                const uint32_t tab = l_tab[o];
                const uint32_t terr = tab >> 8;
                uint32_t *cell = l_state + k * lstride + (tab & 0xff);
                uint32_t obj = *cell;
                if (STEP) {
                    // ref:376-379: every SOUP lying on a tile that is cooking ticks; ref:387-391: a finished kitchen starts empty
                    if ((obj & 0xff) == O_SOUP && np_tick(obj) >= 0 && np_tick(obj) < l_rtime[recipe_index(obj)]) obj += 0x01000000u;
                    if (l_done[k]) obj = 0;
                    *(active ? cell : l_dump) = obj;
                }
                // the ten viewer-independent bytes: channels shift + 0..9
                const uint32_t name = obj & 0xff;
                const bool inpot = name == O_SOUP && terr == T_POT;
                // ref:165-171 terrain one-hot (POT, COUNTER, ONION_SRC, DISH_SRC | SERVING, TOMATO_SRC)
                uint32_t c0 = (terr >= 1 && terr <= 4) ? 1u << (8 * (terr - 1)) : 0u;
                uint32_t c1 = (terr >= 5 && terr <= 6) ? 1u << (8 * (terr - 5)) : 0u;
                // ref:182-203 a soup in a pot: onions on channel shift + 5, tick on shift + 6; anywhere else: the soup flag, shift + 7
                if (inpot) c1 |= (((obj >> 8) & 0xff) << 8) | (static_cast<uint32_t>(max(np_tick(obj), 0)) << 16);
                if (name == O_SOUP && !inpot) c1 |= 1u << 24;
                const uint32_t c2 = (name == O_DISH ? 1u : 0u) | (name == O_ONION ? 0x100u : 0u);   // ref:205-211
                // the tile's C bytes: 5P zero bytes (the players' -- poked below), then the ten
                uint32_t w[(C + 3) / 4];
#pragma unroll
                for (int j = 0; j < (C + 3) / 4; ++j) w[j] = 0;
                // byte `shift` sits at bit 8 * (shift & 3) of dword shift / 4
                constexpr int d0 = shift / 4, sh = 8 * (shift & 3);
                if (sh == 0) { w[d0] = c0; w[d0 + 1] = c1; w[d0 + 2] = c2; }
                else {
                    w[d0] = c0 << sh;
                    w[d0 + 1] = (c0 >> (32 - sh)) | (c1 << sh);
                    w[d0 + 2] = (c1 >> (32 - sh)) | (c2 << sh);
                    if (d0 + 3 < (C + 3) / 4) w[d0 + 3] = c2 >> (32 - sh);
                }
#pragma unroll
                for (int v = 0; v < P; ++v) {
                    int8_t *b = active ? l_stage + v * vstride + fw * C : reinterpret_cast<int8_t *>(l_dump);
                    if (!active) {                                     // (a dump of 64 bytes takes any tile)
#pragma unroll
                        for (int j = 0; j < C / 4; ++j) reinterpret_cast<uint32_t *>(b)[j] = w[j];
                        continue;
                    }
                    typedef uint32_t __attribute__((aligned(1))) u32_any;
                    typedef uint16_t __attribute__((aligned(1))) u16_any;
#pragma unroll
                    for (int j = 0; j < C / 4; ++j) reinterpret_cast<u32_any *>(b)[j] = w[j];
                    if (C % 4 >= 2) *reinterpret_cast<u16_any *>(b + (C & ~3)) = static_cast<uint16_t>(w[C / 4]);
                    if (C % 2) b[C - 1] = static_cast<int8_t>(w[C / 4] >> (8 * ((C - 1) & 3)));
                }
            }
            wave_sync();
            // ---- pokes: lane = (kitchen, player)
            if (pk_flat >= f0 && pk_flat < f0 + wtiles) {
                int8_t *tile = l_stage + (pk_flat - f0) * C;
#pragma unroll
                for (int v = 0; v < P; ++v) {
                    const int slot = pk_i == v ? 0 : (pk_i < v ? pk_i + 1 : pk_i);          // ref:221-234: the viewer first, the others in index order
                    int8_t *b = tile + v * vstride;
                    b[slot] = 1;
                    b[P + 4 * slot + pk_ori] = 1;
                    if (pk_flag) b[pk_flag] = 1;
                }
            }
            wave_sync();
            if (f0 == 0 && step == 0) land_prefetch();                 // the next group's loads: waited for here, in front of this group's first store
            // ---- flush: contiguous bytes of every viewer's plane
            const int nbytes = wtiles * C;
#pragma unroll
            for (int v = 0; v < P; ++v) {
                int8_t *dst = step_obs + v * a.obs_player_stride + plane0 + static_cast<int64_t>(f0) * C;
                const int8_t *src = l_stage + v * vstride;
                // 16-byte stores need a dword-aligned destination only (store16).  A plane that starts off a dword (odd C -- one or
                // three players -- behind an odd number of kitchens x tiles) gets its first 1..3 bytes singly and the rest as 16-byte
                // stores of LDS reads at that byte offset (LDS reads any alignment, tools/probes/lds_unaligned.hip)
                const int head = min(nbytes, static_cast<int>(-reinterpret_cast<uintptr_t>(dst) & 3));   // (wave-uniform)
                if (lane < head) dst[lane] = src[lane];
                const int n16 = (nbytes - head) >> 4;
                if (head == 0) {
                    // (chunk by chunk: four chunks per lane read first and then stored -- one LDS latency per 4 KB -- measured slower,
                    // 64.5 against 60.6 us at 3 x 65 536 kitchens: the paced stores beat the bursts, as in the 2-player kernel)
                    for (int c = lane; c < n16; c += 64) store16<SM>(dst + 16 * c, reinterpret_cast<const uint4 *>(src)[c]);
                } else {
                    typedef uint32_t __attribute__((aligned(1))) u32_any;
                    for (int c = lane; c < n16; c += 64) {
                        const u32_any *q = reinterpret_cast<const u32_any *>(src + head + 16 * c);
                        store16<SM>(dst + head + 16 * c, make_uint4(q[0], q[1], q[2], q[3]));
                    }
                }
                for (int c = head + (n16 << 4) + lane; c < nbytes; c += 64) dst[c] = src[c];
            }
            wave_sync();                                               // the next window overwrites the staging ranges
        }

        // ---- records LDS -> the trajectory's record slot of this step, and after the last step -> HBM (the engine's block: the next
        //      launch's input)
        const bool last_step = step + 1 == n_steps;
        if ((STEP && last_step) || a.rec_out) {
            uint4 *dst = reinterpret_cast<uint4 *>(a.state) + k0 * recq;
            uint4 *rdst = a.rec_out ? reinterpret_cast<uint4 *>(FUSED ? a.rec_out + step * a.rec_step_stride : a.rec_out) + k0 * recq : nullptr;
            for (int c = lane; c < nk * recq; c += 64) {
                const int k = np_div(c, a.inv_recq), j = c - k * recq;
                const uint32_t *sp = l_state + k * lstride + 4 * j;
                const uint4 v = make_uint4(sp[0], sp[1], sp[2], sp[3]);
                if (STEP && last_step) dst[c] = v;
                if (rdst) rdst[c] = v;
            }
        }
        if (FUSED) {
#pragma unroll
            for (int i = 0; i < P; ++i) raw[i] = raw_next[i];
        }
        wave_sync();                                                   // the next step / group overwrites staging and the LDS records
      }   // step loop
    }
}

// every kitchen to the standard start state (ref:387-391)
__global__ void np_fill_start_kernel(uint32_t *state, int64_t n, int S, int P, int recd, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n * recd) return;
    const int d = static_cast<int>(c % recd);
    const uint32_t start[kMaxP] = {p0, p1, p2, p3};
    state[c] = (d >= S && d < S + P) ? start[d - S] : 0u;
}

// per-kitchen episode control (see oc_episode_control)
__global__ void np_episode_control_kernel(uint32_t *state, int64_t n, int S, int P, int recd, int horizon, uint32_t p0, uint32_t p1, uint32_t p2,
                                          uint32_t p3, const int32_t *mask, const int32_t *ts) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t *rec = state + k * recd;
    const uint32_t start[kMaxP] = {p0, p1, p2, p3};
    if (mask && mask[k] != 0)
        for (int d = 0; d < recd; ++d) rec[d] = (d >= S && d < S + P) ? start[d - S] : 0u;
    if (ts) rec[S + 2 * P] = static_cast<uint32_t>(min(max(ts[k], 0), horizon - 1));
}

}  // namespace oc_np
