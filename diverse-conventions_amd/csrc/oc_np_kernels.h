// oc_np_kernels.h -- kitchens with 1, 3 or 4 players (gfx950).
//
// The reference's Python MDP and encoder are N-player (envs/overcooked2_reimplement.py: per-viewer player slots :221-234,
// the dish rule that only pays with exactly two players :261-263; envs/overcooked2_env.py:29 `num_players`; layouts
// multiplayer_schelling and simple_single); its training scripts and the north star of this engine are 2-player, and the
// hand-tuned step kernel (oc_step_body.h) is built around two players (packed positions, 20 channels = 5 dwords per tile,
// the viewer swap).  Other player counts get this separate, plain path: one thread per kitchen for the transition, one
// thread per (viewer, tile) for the encoding staged through LDS so that the plane bytes leave coalesced.  Same semantics,
// same buffers and layouts with P and C = 5P + 10 generalised; bit-exact against fixtures recorded from the reference
// (tests/golden/np_*.npz); no claim on the roofline.  "ref:" = envs/overcooked2_reimplement.py.
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
    const uint32_t *terrain;        // [S]
    const uint32_t *out_tab2;       // [2S]: low byte of entry 2*o = state tile index of output tile o = x*H + y
    const void *actions;            // [P][N] i32 or i64 (NULL: encode only)
    int act64;
    int8_t *obs;                    // viewer v at obs + v * obs_player_stride; [N][W][H][C]
    int64_t obs_player_stride;
    int32_t *reward, *done;
    int64_t n;
    int32_t W, H, S, P, C, recd;
    int32_t horizon, placement_rew, soup_rew;
    int32_t rvalues[16], rtimes[16];
    uint32_t start_pp[kMaxP];
    uint32_t *rec_out;              // encode launch of a step: the records it reads also go here ([N][recd], the trajectory's slot); NULL: not recorded
};

__device__ __forceinline__ int np_delta(int d, int W) { return d == A_NORTH ? -W : d == A_SOUTH ? W : d == A_EAST ? 1 : d == A_WEST ? -1 : 0; }   // ref:22-32
__device__ __forceinline__ int np_tick(uint32_t o) { return static_cast<int32_t>(o) >> 24; }
__device__ __forceinline__ int np_count(uint32_t o) { return static_cast<int>(((o >> 8) & 0xff) + ((o >> 16) & 0xff)); }

// ref:381-385 get_state_transition + the wrappers' done / reward / reset rule (envs/overcooked2_env.py:327-339,
// pantheonrl_extension/vectorenv.py:369-370), one thread per kitchen.
// (Staging the block's records through LDS for coalesced access was measured and is slower: 111 vs 109 us per step at 4 x 65 536
// kitchens -- the index arithmetic of the copy loops and two barriers cost more than the scattered dword accesses, which L2 serves.
// Round 3 tried it once more per wavefront -- 64 kitchens' records as one flat 16-byte-per-lane copy at an odd LDS stride, no barrier:
// 107.5-108.0 against 108.3-109.0 us, i.e. nothing: this kernel's time is not in its record accesses.)
// (Round 4 folded this kernel into the encode kernel -- each block stepping its own four kitchens on the records it holds in LDS anyway,
// one launch per step: bit-exact and TWICE as slow, 218.8 against 108.8 us at 4 x 65 536 kitchens, 155.8 against 84.9 with three
// players.  The transition is a serial chain per kitchen; here 65 536 of them overlap each other, there four threads of a block run it
// while 252 wait, and LDS lets only five blocks share a CU.  Of the step's 101 us of kernel time the encode launch is 79 us at
// 6.4 TB/s of plane stores and this launch 21.6 us, so what is left for this path is the launch boundary, not the bytes.)
__global__ void np_step_kernel(const NpArgs a) {
    const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    uint32_t *rec = a.state + k * a.recd;
    const int S = a.S, P = a.P, W = a.W;
    int pos[kMaxP], ori[kMaxP], act[kMaxP];
    uint32_t held[kMaxP];
    for (int i = 0; i < P; ++i) {
        const uint32_t pp = rec[S + i];
        pos[i] = pp & 0xff; ori[i] = (pp >> 8) & 0xff;
        held[i] = rec[S + P + i];
        const int raw = a.act64 ? static_cast<int>(reinterpret_cast<const long long *>(a.actions)[i * a.n + k])
                                : reinterpret_cast<const int *>(a.actions)[i * a.n + k];
        act[i] = sanitize_action(raw);
    }
    int ts = static_cast<int>(rec[S + 2 * P]);
    // ---- ref:301-354 resolve_interacts.  get_pot_states (ref:272-281) is only consulted by the dish rule, which pays only when
    // there are exactly two players (ref:262-263): with P != 2 a dish pick-up is never rewarded.
    int reward = 0;
    for (int i = 0; i < P; ++i) {
        if (act[i] != A_INTERACT) continue;
        const int ipos = pos[i] + np_delta(ori[i], W);
        const uint32_t terr = a.terrain[ipos];
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
            const int time = a.rtimes[recipe_index(t)];
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
                const int time2 = a.rtimes[recipe_index(t)];
                const bool cooking = np_tick(t) >= 0 && np_tick(t) < time2, ready = np_tick(t) >= 0 && np_tick(t) >= time2;
                if ((t & 0xff) == O_SOUP && !cooking && !ready && np_count(t) > 0 && np_count(t) == 3) t &= 0x00FFFFFFu;
            }
        } else if (terr == T_SERVING && hname == O_SOUP) {                           // ref:350-353, 283-285
            reward += a.rvalues[recipe_index(held[i])];
            held[i] = 0;
        }
        rec[ipos] = t;
    }
    // ---- ref:368-371 resolve_movement, ref:393-399 _move_if_direction, ref:356-366 _handle_collisions
    int npos[kMaxP], nori[kMaxP];
    for (int i = 0; i < P; ++i) {
        if (act[i] == A_INTERACT) { npos[i] = pos[i]; nori[i] = ori[i]; continue; }
        const int cand = pos[i] + np_delta(act[i], W);
        nori[i] = act[i] == A_STAY ? ori[i] : act[i];
        npos[i] = a.terrain[cand] != T_AIR ? pos[i] : cand;
    }
    bool blocked = false;
    for (int x = 0; x < P; ++x)
        for (int y = x + 1; y < P; ++y) blocked = blocked || npos[x] == npos[y] || (npos[x] == pos[y] && pos[x] == npos[y]);
    for (int i = 0; i < P; ++i) {
        if (!blocked) pos[i] = npos[i];
        ori[i] = nori[i];
    }
    // ---- ref:373-379 step_environment_effects
    ts += 1;
    const bool is_done = ts >= a.horizon;                                            // envs/overcooked2_env.py:334
    for (int t = 0; t < S; ++t) {
        uint32_t o = rec[t];
        if (is_done) { if (o) rec[t] = 0; continue; }                                // reset: ref:387-391
        if ((o & 0xff) == O_SOUP && np_tick(o) >= 0 && np_tick(o) < a.rtimes[recipe_index(o)]) rec[t] = o + 0x01000000u;
    }
    for (int i = 0; i < P; ++i) {
        rec[S + i] = is_done ? a.start_pp[i] : (static_cast<uint32_t>(pos[i]) | (static_cast<uint32_t>(ori[i]) << 8));
        rec[S + P + i] = is_done ? 0u : held[i];
        a.reward[i * a.n + k] = reward;                                              // sum, to every player (overcooked2_env.py:336-339)
    }
    rec[S + 2 * P] = is_done ? 0u : static_cast<uint32_t>(ts);
    a.done[k] = is_done ? 1 : 0;
}

// ref:165-259 lossless_state_encoding for every viewer, output order (W, H, C) (envs/overcooked2_env.py:324).
// Block = `kpb` whole kitchens (as many as give at most kNpThreads tiles; an even number when S is odd, so that every block's
// byte range in a viewer's plane starts on a dword whenever C = 5P + 10 is even; the store loop checks the address either way).  The kitchens' records are copied into LDS first (consecutive dwords: coalesced),
// every thread then encodes ONE tile for ALL viewers from there -- the terrain / object / held-object bytes are the same for every
// viewer, only the player slots differ (ref:221-234) -- into one staging range per viewer, and the block writes the P contiguous
// byte ranges out 16 bytes per lane.  (The first version ran one viewer per block straight from global memory: ~11 scattered
// dword loads per thread and viewer, 168 us for 4 x 65 536 kitchens of multiplayer_schelling against 400 MB of planes.)
#ifndef OC_NP_SM
#define OC_NP_SM SM_PLAIN      // store mode of the plane stores (store16)
#endif
constexpr int kNpThreads = 256;
constexpr int kNpMaxC = 5 * kMaxP + 10;             // 30 bytes per tile with four players

__host__ __device__ inline int np_kitchens_per_block(int S) {
    int kpb = kNpThreads / S;
    if (kpb < 1) kpb = 1;                            // (S <= 255 < kNpThreads, so this does not happen)
    if ((S & 1) && kpb > 1) kpb &= ~1;               // an even kitchen count per block: with C = 5P + 10 even (P = 2, 4) every block's bytes are then a
                                                     // multiple of 4; with odd C (P = 1, 3) only of 2 -- the kernel checks each destination and stores bytes
    return kpb;
}
__host__ __device__ inline size_t np_encode_lds_bytes(int S, int P, int recd) {
    const int kpb = np_kitchens_per_block(S), C = 5 * P + 10;
    const size_t per_viewer = (static_cast<size_t>(kpb) * S * C + 15) / 16 * 16;
    return static_cast<size_t>(kpb) * recd * 4 + 16 + per_viewer * P;
}

__global__ __launch_bounds__(kNpThreads) void np_encode_kernel(const NpArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t np_lds[];
    const int S = a.S, P = a.P, C = a.C, shift = 5 * P, recd = a.recd;
    const int kpb = np_kitchens_per_block(S);
    const int64_t k0 = static_cast<int64_t>(blockIdx.x) * kpb;
    const int nk = static_cast<int>(min(static_cast<int64_t>(kpb), a.n - k0));
    const int ntiles = nk * S;
    uint32_t *recs = np_lds;                                                   // [kpb][recd]
    const int rec_dwords = (kpb * recd + 3) & ~3;
    int8_t *stage = reinterpret_cast<int8_t *>(np_lds + rec_dwords);           // [P][per_viewer]
    const int per_viewer = (kpb * S * C + 15) / 16 * 16;
    // records -> LDS, staging ranges -> 0
    // (a recorded step: the post-step records this block reads anyway leave for the trajectory's record slot from here -- the same
    // coalesced dwords, no copy pass over the state block behind the step)
    for (int i = threadIdx.x; i < nk * recd; i += kNpThreads) {
        const uint32_t w = a.state[k0 * recd + i];
        recs[i] = w;
        if (a.rec_out) a.rec_out[k0 * recd + i] = w;
    }
    for (int i = threadIdx.x; i < per_viewer * P / 16; i += kNpThreads) reinterpret_cast<uint4 *>(stage)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if (threadIdx.x < ntiles) {
        const int k = threadIdx.x / S, o = threadIdx.x - k * S, t = static_cast<int>(a.out_tab2[2 * o] & 0xff);
        const uint32_t *rec = recs + k * recd;
        const uint32_t terr = a.terrain[t], obj = rec[t];
        // viewer-independent bytes of this tile: (channel, value) pairs, at most four
        int ch[4], val[4], nb = 0;
        if (terr > T_AIR) { ch[nb] = shift + static_cast<int>(terr) - 1; val[nb++] = 1; }           // ref:165-171
        const uint32_t name = obj & 0xff;
        bool soup = false, dish = false, onion = false;
        if (name == O_SOUP) {                                                                      // ref:182-203
            if (terr == T_POT) {
                ch[nb] = shift + 5; val[nb++] = static_cast<int>((obj >> 8) & 0xff);
                ch[nb] = shift + 6; val[nb++] = max(np_tick(obj), 0);
            } else soup = true;
        } else if (name == O_DISH) dish = true;                                                    // ref:205-207
        else if (name == O_ONION) onion = true;                                                    // ref:209-211
        // players standing here (ref:220-257): their held objects are viewer-independent too
        uint32_t here = 0;                                                                         // bit i: player i on this tile
        uint32_t oris = 0;                                                                         // 2 bits per player
        for (int i = 0; i < P; ++i) {
            const uint32_t pp = rec[S + i];
            if (static_cast<int>(pp & 0xff) != t) continue;
            here |= 1u << i;
            oris |= ((pp >> 8) & 3u) << (2 * i);
            const uint32_t h = rec[S + P + i] & 0xff;
            soup |= h == O_SOUP; dish |= h == O_DISH; onion |= h == O_ONION;
        }
        for (int v = 0; v < P; ++v) {
            int8_t *b = stage + v * per_viewer + threadIdx.x * C;                                  // this tile's C bytes, already zero
            for (int j = 0; j < nb; ++j) b[ch[j]] = static_cast<int8_t>(val[j]);
            if (soup) b[shift + 7] = 1;
            if (dish) b[shift + 8] = 1;
            if (onion) b[shift + 9] = 1;
            if (here) {
                int other = 1;
                for (int i = 0; i < P; ++i) {
                    const int slot = i == v ? 0 : other++;
                    if (!(here & (1u << i))) continue;
                    b[slot] = 1;
                    b[P + 4 * slot + ((oris >> (2 * i)) & 3)] = 1;
                }
            }
        }
    }
    __syncthreads();
    // every viewer's range is contiguous in its plane and starts on a dword (16-byte stores need no more on gfx950, see store16)
    const int nbytes = ntiles * C;
    for (int v = 0; v < P; ++v) {
        int8_t *dst = a.obs + v * a.obs_player_stride + k0 * S * C;
        const int8_t *src = stage + v * per_viewer;
        int done_bytes = 0;
        if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
            const int n16 = nbytes >> 4;
            for (int i = threadIdx.x; i < n16; i += kNpThreads) store16<OC_NP_SM>(dst + 16 * i, reinterpret_cast<const uint4 *>(src)[i]);
            done_bytes = n16 << 4;
        }
        for (int i = done_bytes + threadIdx.x; i < nbytes; i += kNpThreads) dst[i] = src[i];
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
