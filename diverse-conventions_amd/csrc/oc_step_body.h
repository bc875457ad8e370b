// oc_step_body.h -- the step + encode kernel body for ONE layout specialisation.  No include guard: a multi-layout
// module includes it once per layout, each time with its own OC_NS (namespace inside oc_dev) and, under OC_JIT,
// its own OC_S / OC_KPW / ... constants (see the KP_* macros in oc_step_common.h, expanded where they are used).
// See overcooked_engine.hip for the execution model.
#ifndef OC_NS
#error "define OC_NS (the namespace of this specialisation) before including oc_step_body.h"
#endif

// kitchens a group OWNS (OC_KPW sizes LDS and the state registers: one kitchen more under OC_HALO_R)
#undef KP_group
#if defined(OC_GROUP_K)
#define KP_group(a) (OC_GROUP_K)
#else
#define KP_group(a) KP_kpw(a)
#endif

namespace oc_dev {
namespace OC_NS {

struct LdsView {
    uint32_t *terrain, *pots, *counters, *ilut;
    uint2 *out_tab2;
    int32_t *rvalues, *rtimes;
    uint32_t *state;
    uint32_t *kd;               // per-kitchen derived block, 8 dwords: pos0|pos1<<8|done<<16, oh0, oh1, hf0, hf1
    uint32_t *dump;             // 16 dwords nobody reads: where idle lanes' stores go (cheaper than branching)
    uint32_t *stage[2];
};

__device__ __forceinline__ LdsView carve(uint32_t *base, const StepArgs &a) {
    LdsView v;
    v.out_tab2 = reinterpret_cast<uint2 *>(base);                 // 8-byte aligned pairs
    v.terrain = base + 2 * KP_S(a);
    v.rvalues = reinterpret_cast<int32_t *>(base + 3 * KP_S(a));
    v.rtimes = v.rvalues + 16;
    v.ilut = base + 3 * KP_S(a) + 32;
    v.pots = v.ilut + kLutSize;
    v.counters = v.pots + KP_n_pots(a);
    uint32_t *p = base + lds_table_dwords(KP_S(a), KP_n_pots(a), KP_n_counters(a));
    v.kd = p;                                                      // 16-byte aligned rows
    p += KP_kpw(a) * kKd;
    v.dump = p;
    p += 16;
    v.state = p;
    p += align4(KP_kpw(a) * KP_lstride(a));
    v.stage[0] = p;
#if defined(OC_HALO_R) && OC_HALO_R > 0
    v.stage[1] = p + (KP_gpasses(a) * kWave + 16) * kTileDwords;   // (16 tiles in front of each viewer's passes: what one flush leaves to the next, see step_body)
#else
    v.stage[1] = p + KP_gpasses(a) * kWave * kTileDwords;
#endif
    return v;
}

__device__ __forceinline__ int cook_time(const LdsView &L, const StepArgs &a, uint32_t obj) {
    if (KP_uniform_time(a) >= 0) return KP_uniform_time(a);
    return L.rtimes[(4 * ((obj >> 8) & 0xff) + ((obj >> 16) & 0xff)) & 15];   // ref:56-57,156-157
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
        for (int i = 0; i < KP_n_counters(a); ++i) lying |= ((rec[L.counters[i]] & 0xff) == O_DISH);
        if (!lying) rew = KP_dish_rew(a);
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

// One launch = one env step for every kitchen (STEP) or a pure featurisation of the given
// state block (!STEP).  VEC16: every viewer plane base is 16-byte aligned.
// block_id / n_blocks: this wavefront's index among the wavefronts that share the work described by `a`, and
// their number -- blockIdx.x / gridDim.x for a launch over one engine, a sub-range of the grid when one launch
// serves several engines (oc_group_step).
// BATCH (!STEP only): `a` describes batch_items blocks of n_kitchens records each, item b at state +
// b * state_batch_stride (uint4) with its planes at obs + b * obs_batch_stride (bytes): a whole gathered trajectory
// of one peer re-featurised in one launch.
// lds_dword_offset / lane_id: where the body's LDS block starts inside the workgroup's dynamic LDS, and the lane index, when it
// runs as ONE wavefront of a larger workgroup (the policy kernel folds the env step of a kitchen group into its own launch,
// policy_ops.hip); a launch of its own leaves them out.  An OFFSET, not a pointer: a generic pointer would turn every LDS access
// into a flat instruction, which counts on vmcnt as well -- each LDS read then waits for all plane stores in flight (measured:
// 2 us per flush group instead of 0.2).
// COHERENT: the action buffer was filled earlier in the same launch (see load_action).
// TWOPTR: the two players' actions arrive as two [N] buffers (a.actions, a.actions1) instead of one [P][N] tensor -- the drop-in
// classes hand the ego's and the partner's action tensors over as they are (oc_step_players), no assembly pass in front of the step.
template <bool STEP, bool ACT64, bool VEC16, int QN, int SM = SM_PLAIN, bool FUSED = false, bool BATCH = false, bool COHERENT = false, bool TWOPTR = false>
__device__ __forceinline__ void step_body(const StepArgs &a, const int block_id, const int n_blocks, int lds_dword_offset = 0, int lane_id = -1) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    uint32_t *const lds = lds_all + lds_dword_offset;
    const LdsView L = carve(lds, a);
    const int lane = lane_id >= 0 ? lane_id : static_cast<int>(threadIdx.x);
    const int S = KP_S(a);

    // ---- software pipeline: registers hold the *next* group's state block and actions
    // Loads are unconditional (indices clamped into the arrays) so that they land directly in the
    // pipeline registers: a predicated load would be copied -- and therefore waited for -- on the spot.
    // (one 128-bit vector value per chunk, not a struct of four words: the landing below then ties ONE register tuple per chunk.  As
    // four separately tied words -- rounds 1-3 -- hipcc 7.2 dropped the copies of two of them in one build of the generic kernel and
    // the second group of a wavefront wrote stale words into its last two kitchens: found in round 4, when the whole parity file first
    // ran on that kernel at a size where wavefronts take two groups)
    u32x4 pre[QN];
    int pa0, pa1;
    const int64_t last_chunk = KP_n_kitchens(a) * KP_recq(a) - 1;
    // group -> (batch item, first kitchen of the group inside the item); one item unless BATCH
    auto item_of = [&](int g) { return BATCH ? g / a.batch_groups : 0; };
    // ---- what a group is.  Ordinarily KPW consecutive kitchens, whose tiles -- and hence plane bytes -- are the group's own.
    // Under OC_XW_SUM (specialised build, one launch per step, one engine) a group is sized for the XCD that runs it: the XCDs do
    // not stream stores at the same rate -- on equal shares the odd ones end 1-2 us after the even ones and the launch is as long as
    // the slowest (tools/probes/ticket_atomic.hip; per-XCC end times of tools/trace_phases.py) -- and workgroup b runs on XCD
    // b mod 8, so group g, taken by a workgroup of XCD g mod 8 whenever the grid is a multiple of eight, *owns* byte (g mod 8) of
    // OC_XW_PACK kitchens, eight consecutive groups covering OC_XW_SUM.  A wavefront's time follows its own bytes, so the sizes must
    // move by single kitchens; but a group whose first plane byte is not 64-byte aligned pays a read-modify-write for every partial
    // 64-byte word of every store instruction (17 / 15 kitchens per group with cramped_room's 400-byte kitchens: 18.5 us against
    // 11.9).  So ownership of *kitchens* (transition, state, reward, done) and of *plane bytes* are separated: the byte range of a
    // group runs from tile T(k0) to tile T(k1), T(k) = k*S rounded DOWN to a multiple of 16 tiles = 320 bytes (T(N) = N*S), i.e. the
    // first tiles of its range may belong to kitchen k0-1 -- the *halo* kitchen, which the group loads and steps as well (the
    // transition runs one kitchen per lane with most lanes idle: the duplicate costs nothing) without writing its state, reward or
    // done -- and the last tiles of its own last kitchen are written by the next group.  It still ENCODES through the end of its
    // last kitchen: phase E is also where lying soups tick in the state.
    // Correct under any block -> XCD mapping and any grid: a group's kitchens and bytes are a function of its index alone.
#if defined(OC_XW_SUM)
    constexpr bool kXW = !FUSED && !BATCH;
#else
    constexpr bool kXW = false;
#endif
    // ---- OC_HALO_R (specialised build of the module for fresh destinations, for an engine whose planes can start off a 64-byte
    // boundary: a kitchen count whose planes are no multiple of 64 bytes).  A 64-byte word of memory written by two store
    // instructions -- let alone two wavefronts -- costs a read-modify-write each time, and with a plane starting 16 / 32 / 48 bytes
    // off, every 1 KB store instruction of its flushes splits two words (65 535 kitchens of cramped_room into time-major slots: 18.3 us
    // per launch against 11.9 at 65 536; there viewer 0's plane is 0 or 32 bytes off depending on the slot, viewer 1's 48 or 16).
    // A plane's bytes are therefore owned in ranges that start r tiles EARLY, 20 r = the plane's offset mod 64 (r = 0, 4, 8, 12; read
    // off the pointers, per viewer and launch): a group of OC_GROUP_K kitchens [k0, k1) writes viewer v's tiles [k0 S - r_v, k1 S - r_v)
    // -- the last r_v tiles of kitchen k0 - 1, which it loads and steps as a halo kitchen like the per-XCD experiment above, and not
    // its own last r_v (the next group's).  Passes start at tile k0 S - R, R = max r_v, so the flushes of the viewer with r_v = R begin
    // and end on 64-byte boundaries; the other viewer's lag 16 - (R - r_v) tiles behind the passes, the tiles a flush leaves over
    // waiting in a 16-tile area in front of its staged passes.  The plane's first group starts at tile 0 and its last group ends at
    // N S: two groups of a launch keep split words.  Correct for any plane address: only the alignment depends on it.
    // Streaming stores only: plain stores (fresh slots that a policy launch reads next) merge their partial words in L2, and the halo
    // kitchen, the extra tiles and the sixth pass of a 16-kitchen group are then pure cost -- collecting with the policies in the
    // loop at 65 535 kitchens 51.6 against 49.8 us per step.
#if defined(OC_HALO_R) && OC_HALO_R > 0 && !defined(OC_XW_SUM)
    constexpr bool kHR = !FUSED && !BATCH && SM != SM_PLAIN;
    constexpr int kCarry = kHR ? 16 * kTileDwords : 0;
#else
    constexpr bool kHR = false;
    constexpr int kCarry = 0;
#endif
    // A halo kitchen is READ by a group that does not own it, in the launch in which its owner -- another wavefront -- writes its
    // post-step record: the launch therefore reads every record from a.state and writes it to a.state_out, a second block that the
    // host swaps with the first after the launch (launch() in overcooked_engine.hip).  Without one (a launch being captured into a
    // graph, whose replays could not swap) there are no halo kitchens: r = 0, the lane-shifted flush loop below.
    int hr_r[2] = {0, 0}, hr_R = 0;                  // (wave-uniform)
    if (kHR && a.state_out) {
        const uint32_t m0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(a.obs)) & 63u;
        const uint32_t m1 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(a.obs + a.obs_player_stride)) & 63u;
        if (((m0 | m1) & 15u) == 0) { hr_r[0] = static_cast<int>(m0 >> 4) * 4; hr_r[1] = static_cast<int>(m1 >> 4) * 4; }
        hr_R = max(hr_r[0], hr_r[1]);
    }
    struct Geom {
        int64_t k0;          // first kitchen in LDS (the halo kitchen if there is one)
        int n_lds, halo;     // kitchens in LDS; how many of them (0 / 1), in front, are not the group's own
        int64_t tile0;       // first tile of the group's byte range, flat index kitchen * S + tile
        int ntiles, nflush;  // tiles to encode from tile0 on; tiles whose bytes the group writes
        bool first, last;    // (OC_HALO_R) the plane's first / last group
    };
    auto geom_of = [&](int g, int item) {
        Geom G;
        int64_t k0;
        int nk;
#if defined(OC_XW_SUM)
        if (kXW) {
            k0 = static_cast<int64_t>(g >> 3) * OC_XW_SUM + static_cast<int64_t>((OC_XW_PREFIX >> (8 * (g & 7))) & 0xFFull);
            nk = static_cast<int>((OC_XW_PACK >> (8 * (g & 7))) & 0xFFull);
        } else
#endif
        {
            k0 = static_cast<int64_t>(BATCH ? g - item * a.batch_groups : g) * KP_group(a);
            nk = KP_group(a);
        }
        nk = static_cast<int>(max(static_cast<int64_t>(0), min(static_cast<int64_t>(nk), KP_n_kitchens(a) - k0)));
        G.first = false; G.last = false;
        if (kHR && nk > 0) {
            G.first = k0 == 0; G.last = k0 + nk >= KP_n_kitchens(a);
            G.halo = (G.first || hr_R == 0) ? 0 : 1;
            G.k0 = k0 - G.halo; G.n_lds = nk + G.halo;
            G.tile0 = G.first ? 0 : k0 * S - hr_R;
            G.ntiles = G.nflush = static_cast<int>((k0 + nk) * S - G.tile0);
        } else if (kXW && nk > 0) {
            const int64_t k1 = k0 + nk;
            const int64_t t_lo = (k0 * S) & ~static_cast<int64_t>(15);
            const int64_t t_hi = k1 >= KP_n_kitchens(a) ? KP_n_kitchens(a) * S : (k1 * S) & ~static_cast<int64_t>(15);
            G.halo = t_lo < k0 * S ? 1 : 0;                  // (S >= 16, checked by the host: the halo is one kitchen at most)
            G.k0 = k0 - G.halo; G.n_lds = nk + G.halo;
            G.tile0 = t_lo; G.ntiles = static_cast<int>(k1 * S - t_lo); G.nflush = static_cast<int>(t_hi - t_lo);
        } else {
            G.halo = 0; G.k0 = k0; G.n_lds = nk;
            G.tile0 = k0 * S; G.ntiles = G.nflush = nk * S;
        }
        return G;
    };
    auto prefetch = [&](int g) {
        const int item = item_of(g);
        const int64_t k0 = geom_of(g, item).k0;
        const int64_t c0 = k0 * KP_recq(a) + lane;
        const u32x4 *st = reinterpret_cast<const u32x4 *>(a.state + (BATCH ? item * a.state_batch_stride : 0));
#pragma unroll
        for (int q = 0; q < QN; ++q) pre[q] = st[min(c0 + q * kWave, last_chunk)];
        if (STEP) {
            const int64_t kid = min(k0 + lane, KP_n_kitchens(a) - 1);
            if (FUSED && !a.actions) {
                pa0 = static_cast<int>(oc_hash_action(a.rng_seed, static_cast<uint32_t>(kid), a.rng_t0, 0));
                pa1 = static_cast<int>(oc_hash_action(a.rng_seed, static_cast<uint32_t>(kid), a.rng_t0, 1));
            } else {
                pa0 = load_action<ACT64, COHERENT>(a.actions, kid);
                pa1 = TWOPTR ? load_action<ACT64, COHERENT>(a.actions1, kid) : load_action<ACT64, COHERENT>(a.actions, KP_n_kitchens(a) + kid);
            }
        } else {
            pa0 = pa1 = A_STAY;
        }
    };
    // Make the compiler's vmcnt bookkeeping see the prefetch registers as landed: vmcnt retires in
    // order, so a wait for these loads placed after plane stores have been issued also waits for the stores.
    auto land_prefetch = [&]() {
#pragma unroll
        for (int q = 0; q < QN; ++q) asm volatile("" : "+v"(pre[q]));
        asm volatile("" : "+v"(pa0), "+v"(pa1));
    };
    unsigned long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (KP_trace(a)) stamp[0] = __builtin_amdgcn_s_memrealtime();
    // ---- tables -> LDS, once per wave: the host lays the table image out exactly as carve() does, so it
    // is one flat 16-byte copy (all loads in flight together instead of one round trip per table).  The table loads go out FIRST
    // (L2 hits), the first group's state and actions (HBM) behind them: vmcnt retires in order, so the counted wait for the table
    // does not wait for the state.  The loads are unconditional and "landed" through an empty asm: left to itself the compiler sinks
    // each into the `if` of its LDS write -- load, vmcnt(0), write, load, vmcnt(0), write: two round trips, the first of them behind
    // the state loads, in front of everything a wavefront does (seen in the code object of the specialised build).
#if defined(OC_XCC_ROT) && OC_XCC_ROT
    int group = (block_id + OC_XCC_ROT) % n_blocks;               // experiment: the groups of XCD x handled by XCD x-1
#else
    int group = block_id;
#endif
    {
        const u32x4 *img = reinterpret_cast<const u32x4 *>(a.tabs);
        const int nq4 = lds_table_dwords(S, KP_n_pots(a), KP_n_counters(a)) / 4;
        u32x4 tq[kTabQ];
#pragma unroll
        for (int q = 0; q < kTabQ; ++q) tq[q] = img[min(lane + q * kWave, nq4 - 1)];
        prefetch(min(group, KP_n_groups(a) - 1));                 // the first group's loads: in flight during the table setup
#pragma unroll
        for (int q = 0; q < kTabQ; ++q) asm volatile("" : "+v"(tq[q]));
#pragma unroll
        for (int q = 0; q < kTabQ; ++q)
            if (lane + q * kWave < nq4) reinterpret_cast<u32x4 *>(lds)[lane + q * kWave] = tq[q];
        for (int c = lane + kTabQ * kWave; c < nq4; c += kWave) reinterpret_cast<u32x4 *>(lds)[c] = img[c];   // huge layouts only
    }

    // where this lane's q-th 16-byte chunk of a group's state block lives in LDS (group independent)
    int loff[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const int c = lane + q * kWave;
        const int k = c / KP_recq(a);
        loff[q] = k * KP_lstride(a) + 4 * (c - k * KP_recq(a));
    }
    // tile walk constants: pass p covers flat tiles p*64 + lane, flat = kitchen*S + output tile
    // flat mode: 64 tiles per pass, tiles of a kitchen may straddle passes; kitchen-aligned mode (kpp > 0):
    // kpp whole kitchens per pass, so a lane keeps its tile for the whole launch (its table entry and all
    // its addresses are loop-invariant) at the price of 64 - kpp*S idle lanes
    const int tpp = KP_kpp(a) > 0 ? KP_kpp(a) * S : kWave;            // tiles per pass
    const int k_first = lane / S, o_first = lane - k_first * S;
    const int dk = KP_kpp(a) > 0 ? KP_kpp(a) : kWave / S, dq = KP_kpp(a) > 0 ? 0 : kWave - dk * S;

    const uint2 tab_fixed = L.out_tab2[o_first];     // kitchen-aligned mode: this lane's tile never changes
    land_prefetch();
    if (KP_trace(a)) stamp[1] = __builtin_amdgcn_s_memrealtime();

    for (; group < KP_n_groups(a); group += n_blocks) {
        const int item = item_of(group);
        const Geom G = geom_of(group, item);
        const int64_t k0 = G.k0;                     // first kitchen in LDS
        const int n_valid = G.n_lds;                 // kitchens in LDS: all of them are stepped, the group's own are written back
        const int halo = (kXW || kHR) ? G.halo : 0;
        const int nq = n_valid * KP_recq(a);

        // ---- prefetched state block -> LDS, then start fetching the next group
#pragma unroll
        for (int q = 0; q < QN; ++q)
            if (lane + q * kWave < nq) {
                uint32_t *d = L.state + loff[q];
                d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
            }
        int raw0 = pa0, raw1 = pa1;
        if (group + n_blocks < KP_n_groups(a)) prefetch(group + n_blocks);   // (wave-uniform; the last group: nothing to fetch)
        wave_sync();
        if (KP_trace(a) && stamp[6] == 0) stamp[6] = __builtin_amdgcn_s_memrealtime();
        const int64_t plane_base = G.tile0 * kChannels;  // byte offset of this group's first tile in a viewer plane
        const int ntiles = G.ntiles, nflush = G.nflush;
        const int npass = (ntiles + tpp - 1) / tpp;

        // FUSED: the whole rollout of this group happens here, its state never leaves LDS between steps
        for (int step = 0; step < (FUSED ? a.n_steps : 1); ++step) {
            const int a0 = sanitize_action(raw0), a1 = sanitize_action(raw1);
            int8_t *const obs_t = a.obs + (FUSED ? step * a.obs_step_stride : 0) + (BATCH ? item * a.obs_batch_stride : 0);
            int32_t *const reward_t = a.reward + (FUSED ? step * a.reward_step_stride : 0);
            int32_t *const done_t = a.done + (FUSED ? step * a.done_step_stride : 0);
            if (FUSED && step + 1 < a.n_steps) {             // next step's actions, in flight during this step
                const int64_t kid = min(k0 + lane, KP_n_kitchens(a) - 1);
                if (!a.actions) {
                    raw0 = static_cast<int>(oc_hash_action(a.rng_seed, static_cast<uint32_t>(kid), a.rng_t0 + step + 1, 0));
                    raw1 = static_cast<int>(oc_hash_action(a.rng_seed, static_cast<uint32_t>(kid), a.rng_t0 + step + 1, 1));
                } else {
                    const int64_t slice = static_cast<int64_t>((step + 1) % a.tape_steps) * a.action_step_stride;
                    raw0 = load_action<ACT64>(a.actions, slice + kid);
                    raw1 = load_action<ACT64>(a.actions, slice + KP_n_kitchens(a) + kid);
                }
            }

            // ---- phase T: lane = kitchen.  ref:381-385 get_state_transition + wrappers' done/reset rule
            if (lane < n_valid) {
                uint32_t *rec = L.state + lane * KP_lstride(a);
                uint32_t posor = rec[S], h0 = rec[S + 1], h1 = rec[S + 2];
                uint32_t is_done = 0;
                if (STEP && !(KP_ablate(a) & 1)) {
                    const int64_t kid = k0 + lane;
                    int ts = static_cast<int>(rec[S + 3]);
                    int pos0 = posor & 0xff, or0 = (posor >> 8) & 0xff, pos1 = (posor >> 16) & 0xff, or1 = posor >> 24;

                    // round 1 of LDS reads happened above (posor, hands, timestep); round 2: everything whose
                    // address depends only on positions/orientations/actions, issued together
                    const int ipos0 = pos0 + dir_delta(or0, KP_delta_pack(a)), ipos1 = pos1 + dir_delta(or1, KP_delta_pack(a));
                    const int cand0 = pos0 + dir_delta(a0, KP_delta_pack(a)), cand1 = pos1 + dir_delta(a1, KP_delta_pack(a));   // STAY/INTERACT: delta 0
                    const uint32_t terr_i0 = L.terrain[ipos0], terr_i1 = L.terrain[ipos1];
                    const uint32_t terr_c0 = L.terrain[cand0], terr_c1 = L.terrain[cand1];
                    uint32_t tile0 = rec[ipos0], tile1 = rec[ipos1];
                    const int rv0 = L.rvalues[recipe_index(h0)], rv1 = L.rvalues[recipe_index(h1)];
                    // ref:272-281 get_pot_states, taken once before anybody acts (ref:302); the first four pots
                    // are read unconditionally (index 0 when absent) so the reads go out together
                    auto nonempty_pot = [](uint32_t o) {
                        const int tick = static_cast<int32_t>(o) >> 24;
                        const int n = static_cast<int>(((o >> 8) & 0xff) + ((o >> 16) & 0xff));
                        return ((o & 0xff) != O_NONE && (tick >= 0 || n < 3)) ? 1 : 0;
                    };
                    const uint32_t po0 = rec[KP_pots4(a, 0)], po1 = rec[KP_pots4(a, 1)], po2 = rec[KP_pots4(a, 2)], po3 = rec[KP_pots4(a, 3)];
                    int pot_states = (KP_n_pots(a) > 0 ? nonempty_pot(po0) : 0) + (KP_n_pots(a) > 1 ? nonempty_pot(po1) : 0) +
                                     (KP_n_pots(a) > 2 ? nonempty_pot(po2) : 0) + (KP_n_pots(a) > 3 ? nonempty_pot(po3) : 0);
                    for (int i = 4; i < KP_n_pots(a); ++i) pot_states += nonempty_pot(rec[L.pots[i]]);
                    if (KP_trace(a) && stamp[7] == 0) stamp[7] = __builtin_amdgcn_s_memrealtime() + (pot_states & 0) + (tile0 & 0) + (tile1 & 0) + (terr_c0 & 0) + (terr_c1 & 0);
                    // interacts in player order on pre-move positions; player 1 sees player 0's effects
                    int rew = interact(a0 == A_INTERACT, terr_i0, tile0, h0, h1, rv0, pot_states, rec, L, a);
                    rec[ipos0] = tile0;
                    if (ipos1 == ipos0) tile1 = tile0;
                    rew += interact(a1 == A_INTERACT, terr_i1, tile1, h1, h0, rv1, pot_states, rec, L, a);
                    rec[ipos1] = tile1;

                    if (KP_trace(a) && stamp[8] == 0) stamp[8] = __builtin_amdgcn_s_memrealtime() + (rew & 0);
                    // ref:393-399 _move_if_direction, ref:356-366 _handle_collisions
                    const int no0 = (a0 >= A_STAY) ? or0 : a0, no1 = (a1 >= A_STAY) ? or1 : a1;
                    const int np0 = (terr_c0 != T_AIR) ? pos0 : cand0;        // cand == pos (own, walkable tile) when not moving
                    const int np1 = (terr_c1 != T_AIR) ? pos1 : cand1;
                    const bool blocked = (np0 == np1) || (np0 == pos1 && pos0 == np1);
                    if (!blocked) { pos0 = np0; pos1 = np1; }
                    or0 = no0; or1 = no1;

                    // ref:373-375 timestep; envs/overcooked2_env.py:334 done; vectorenv.py:369-370 reset rule
                    ts += 1;
                    is_done = ts >= KP_horizon(a) ? 1u : 0u;
                    posor = pos0 | (or0 << 8) | (pos1 << 16) | (static_cast<uint32_t>(or1) << 24);
                    if (is_done) { posor = KP_start_posor(a); h0 = 0; h1 = 0; ts = 0; }
                    rec[S] = posor; rec[S + 1] = h0; rec[S + 2] = h1; rec[S + 3] = static_cast<uint32_t>(ts);
                    if (lane >= halo && !(KP_ablate(a) & 32)) {   // (a halo kitchen's outputs are its owner's; bit 32: diagnostics, timing only)
                        reward_t[kid] = rew;                 // envs/overcooked2_env.py:336-339: sum, to every player
                        reward_t[KP_n_kitchens(a) + kid] = rew;
                        done_t[kid] = static_cast<int32_t>(is_done);
                    }
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
            if (KP_trace(a) && stamp[2] == 0) stamp[2] = __builtin_amdgcn_s_memrealtime();

            // ---- phase E: lane = tile, in output order o = x*H + y per kitchen (envs/overcooked2_env.py:324)
            int k = k_first, o = o_first;
            if (kXW || kHR) {                                // the byte range starts tile0 - k0*S tiles into the LDS kitchens
                const int f = static_cast<int>(G.tile0 - k0 * S) + lane;
                k = f / S; o = f - k * S;
            }
            // One tile of phase E is split in two so that two tiles per lane can be in flight: all LDS
            // reads of both tiles are issued before either tile's arithmetic (the kernel runs at 1-2 waves
            // per SIMD, so instruction-level parallelism inside the wave is what hides LDS latency).
            struct TileIn { uint2 tab; uint4 kq; uint32_t hf1; uint32_t obj; uint32_t *cell; bool active; };
            auto tile_read = [&](int kk, int oo, bool active) {
                TileIn r;
                kk = min(kk, KP_kpw(a) - 1);                     // idle lanes of a ragged last pass read in-range LDS
                r.tab = KP_kpp(a) > 0 ? tab_fixed : L.out_tab2[oo];
                r.kq = *reinterpret_cast<const uint4 *>(L.kd + kk * kKd);       // pp, oh0, oh1, hf0
                r.hf1 = L.kd[kk * kKd + 4];
                uint32_t *cell = L.state + kk * KP_lstride(a) + (r.tab.x & 0xff);
                r.obj = *cell;
                r.cell = active ? cell : L.dump;             // idle lanes store to a dump word instead of branching
                r.active = active;
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
                uint32_t *s0 = r.active ? L.stage[0] + kCarry + slot : L.dump, *s1 = r.active ? L.stage[1] + kCarry + slot : L.dump + 8;
                // bytes: [self here, other here, self orient x4, other orient x4, terrain ...]
                s0[0] = e0 | (e1 << 8) | (A << 16); s0[1] = __builtin_amdgcn_alignbit(B, A, 16); s0[2] = __builtin_amdgcn_alignbit(thi, B, 16); s0[3] = d3; s0[4] = d4;
                s1[0] = e1 | (e0 << 8) | (B << 16); s1[1] = __builtin_amdgcn_alignbit(A, B, 16); s1[2] = __builtin_amdgcn_alignbit(thi, A, 16); s1[3] = d3; s1[4] = d4;
            };
            auto advance = [&]() {
                k += dk; o += dq;
                if (o >= S) { o -= S; ++k; }
            };
            auto encode_passes = [&](int p0, int count) {
                const int pend = min(p0 + count, npass);
                if (KP_ablate(a) & 4) return;
                int p = p0;
#if defined(OC_ENCODE_ILP4) && OC_ENCODE_ILP4
                for (; p + 3 < pend; p += 4) {               // four tiles per lane in flight: what phase E waits for are its LDS round trips
                    const int slot = ((p - p0) * tpp + lane) * kTileDwords;
                    const TileIn r0 = tile_read(k, o, lane < tpp && p * tpp + lane < ntiles);
                    advance();
                    const TileIn r1 = tile_read(k, o, lane < tpp && (p + 1) * tpp + lane < ntiles);
                    advance();
                    const TileIn r2 = tile_read(k, o, lane < tpp && (p + 2) * tpp + lane < ntiles);
                    advance();
                    const TileIn r3 = tile_read(k, o, lane < tpp && (p + 3) * tpp + lane < ntiles);
                    advance();
                    tile_encode(r0, slot);
                    tile_encode(r1, slot + tpp * kTileDwords);
                    tile_encode(r2, slot + 2 * tpp * kTileDwords);
                    tile_encode(r3, slot + 3 * tpp * kTileDwords);
                }
#endif
                for (; p + 1 < pend; p += 2) {               // two tiles per lane in flight
                    const int slot = ((p - p0) * tpp + lane) * kTileDwords;
                    const TileIn r0 = tile_read(k, o, lane < tpp && p * tpp + lane < ntiles);
                    advance();
                    const TileIn r1 = tile_read(k, o, lane < tpp && (p + 1) * tpp + lane < ntiles);
                    advance();
                    tile_encode(r0, slot);
                    tile_encode(r1, slot + tpp * kTileDwords);
                }
                if (p < pend) {
                    const TileIn r0 = tile_read(k, o, lane < tpp && p * tpp + lane < ntiles);
                    advance();
                    tile_encode(r0, ((p - p0) * tpp + lane) * kTileDwords);
                }
            };
            auto flush_passes = [&](int p0, int count) {
                // ---- flush the staged tiles of both viewers: contiguous bytes in the output planes
                const int gbytes = max(0, min(nflush - p0 * tpp, count * tpp)) * kChannels;
                const int64_t goff = plane_base + static_cast<int64_t>(p0) * tpp * kChannels;
#if defined(OC_JIT) && ((defined(OC_FAST_FLUSH) && OC_FAST_FLUSH) || (!defined(OC_FAST_FLUSH) && OC_S <= 32))
                // Specialised build, full flush group (every one but a ragged last): the chunk count is a constant, so all LDS
                // reads of both viewers go out first (one latency instead of one per chunk and viewer in a dependent
                // read / wait / store loop), then the stores, addressed as scalar base + lane offset + immediate.
                // Measured per grid (profiles/r02_ab_geometry_flush.log, 65 536 kitchens): cramped_room 11.93 -> 11.90 us with 16
                // kitchens per group and 11.98 -> 11.80 with 32; coordination_ring 13.88 -> 13.77; counter_circuit 20.00 -> 20.03;
                // asymmetric_advantages 22.09 -> 22.76 (the 24 extra registers cost more there than the loop did): grids of at
                // most 32 tiles only.
                constexpr int kFullBytes = OC_GP * (OC_KPP > 0 ? OC_KPP * OC_S : kWave) * kChannels;
                constexpr int kFullChunks = kFullBytes / 16, kIts = (kFullChunks + kWave - 1) / kWave;
                // (also for ranges that do not start on a 64-byte boundary: this is the in-place path, whose stores the Infinity Cache
                // absorbs -- 65 535 kitchens 11.41 us with it, 11.77 through the lane-shifting loop below)
                if (!kHR && VEC16 && kFullBytes % 16 == 0 && kIts <= 6 && gbytes == kFullBytes) {
                    uint4 r[2][kIts];
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int i = 0; i < kIts; ++i)
                            if ((i + 1) * kWave <= kFullChunks || lane + i * kWave < kFullChunks)
                                r[v][i] = *reinterpret_cast<const uint4 *>(L.stage[v] + 4 * (lane + i * kWave));
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        int8_t *base = obs_t + v * a.obs_player_stride + goff;
#pragma unroll
                        for (int i = 0; i < kIts; ++i)
                            if ((i + 1) * kWave <= kFullChunks || lane + i * kWave < kFullChunks) {
                                if ((i & 3) == 0) store16_s<SM, 0>(base + (i >> 2) * 4096, lane * 16u, r[v][i]);
                                else if ((i & 3) == 1) store16_s<SM, 1024>(base + (i >> 2) * 4096, lane * 16u, r[v][i]);
                                else if ((i & 3) == 2) store16_s<SM, 2048>(base + (i >> 2) * 4096, lane * 16u, r[v][i]);
                                else store16_s<SM, 3072>(base + (i >> 2) * 4096, lane * 16u, r[v][i]);
                            }
                    }
                    return;
                }
#endif
                // (OC_HALO_R: the tiles of this flush, counted from the group's first encoded tile)
                const int t_beg = p0 * tpp, t_end = min((p0 + count) * tpp, ntiles);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    if (KP_ablate(a) & 2) break;
                    int8_t *dst = obs_t + v * a.obs_player_stride + goff;
                    const uint32_t *src = L.stage[v] + kCarry;
                    int vbytes = gbytes;
                    if (kHR) {
                        // this viewer's tiles of the flush: from its own first tile on, `lag` tiles behind the passes until the group's
                        // last flush, which ends r tiles before the last encoded one (the next group's, unless there is none)
                        const int r = hr_r[v];
                        const int lag = (G.first || r == hr_R) ? 0 : 16 - (hr_R - r);
                        const int vend = G.last ? ntiles : ntiles - r;
                        const int fb = p0 == 0 ? (G.first ? 0 : hr_R - r) : t_beg - lag;
                        const int fe = t_end >= ntiles ? vend : min(t_end - lag, vend);
                        vbytes = max(0, fe - fb) * kChannels;
                        src += (fb - t_beg) * kTileDwords;            // (may point into the 16 tiles in front of the passes)
                        dst = obs_t + v * a.obs_player_stride + (G.tile0 + fb) * kChannels;
                    }
                    if (VEC16) {
                        const int nfull = vbytes >> 4;            // whole 16-byte chunks: 1 KiB per wave store
                        // Every store instruction starts on a 64-byte boundary of memory: when the range does not (a viewer plane behind
                        // an odd number of kitchens, a group size that is no multiple of 64 bytes), the first instruction's leading lanes
                        // stay idle instead.  A 64-byte word written by two instructions costs a read-modify-write each time --
                        // 65 535 kitchens of cramped_room (plane 1 starts 48 bytes off) into time-major slots: 18.3 us per launch
                        // against 11.9 at 65 536 (round 4).
                        // (a loop of its own for that case: the aligned one, which is every power-of-two batch, pays nothing for it --
                        // as one loop with the shift as a variable config 4's split ran 2.5 % slower, config 3's launch 3 %)
                        // (streaming stores only: plain and write-through stores into a cached destination merge their partial words
                        // there -- `_pl` at 65 535 kitchens 13.3 us without the shift, 13.8 with it)
                        const int sh = (SM == SM_NT || SM == SM_SC1_NT) ? static_cast<int>((reinterpret_cast<uintptr_t>(dst) >> 4) & 3) : 0;     // (wave-uniform)
                        if (sh == 0) {
                            for (int c = lane; c < nfull; c += kWave) store16<SM>(dst + c * 16, *reinterpret_cast<const uint4 *>(src + 4 * c));
                        } else {
                            for (int c = lane - sh; c < nfull; c += kWave)
                                if (c >= 0) store16<SM>(dst + c * 16, *reinterpret_cast<const uint4 *>(src + 4 * c));
                        }
                        const int rem = (vbytes & 15) >> 2;       // 0..3 dwords, ragged tail of the very last group only
                        if (lane < rem) store4(dst + nfull * 16 + lane * 4, src[nfull * 4 + lane]);
                    } else {
                        for (int c = lane; c * 4 < vbytes; c += kWave) store4(dst + c * 4, src[c]);
                    }
                }
                if (kHR && !G.first && t_end < ntiles) {
                    // the last `lag` tiles of these passes wait for the next flush in front of the staged passes (16-byte chunks: lag
                    // is a multiple of four tiles; LDS operations of a wavefront complete in order, the flush's reads above come first)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const int lag = hr_r[v] == hr_R ? 0 : 16 - (hr_R - hr_r[v]);
                        const uint32_t *from = L.stage[v] + kCarry + (t_end - t_beg - lag) * kTileDwords;
                        uint32_t *to = L.stage[v] + kCarry - lag * kTileDwords;
                        if (lane < lag * kTileDwords / 4) {
                            const uint4 t = *reinterpret_cast<const uint4 *>(from + 4 * lane);
                            *reinterpret_cast<uint4 *>(to + 4 * lane) = t;
                        }
                    }
                }
            };
            // first flush group peeled so that the prefetch lands at a fixed program point: after the first
            // tiles are encoded (latency cover) and before this group's first plane store is issued
#if defined(OC_FIRST_PASSES) && OC_FIRST_PASSES > 0
            const int first_passes = min(OC_FIRST_PASSES, KP_gpasses(a));     // experiment: a shorter first flush group, stores start earlier
#else
            const int first_passes = KP_gpasses(a);
#endif
            encode_passes(0, first_passes);
            wave_sync();
            if (!FUSED || step == 0) land_prefetch();
            flush_passes(0, first_passes);
            wave_sync();
            if (KP_trace(a) && stamp[3] == 0) stamp[3] = __builtin_amdgcn_s_memrealtime();
#if defined(OC_NOUNROLL) && OC_NOUNROLL
#pragma unroll 1
#endif
            for (int p0 = first_passes; p0 < npass; p0 += KP_gpasses(a)) {
                encode_passes(p0, KP_gpasses(a));
                wave_sync();
                flush_passes(p0, KP_gpasses(a));
                wave_sync();
            }
            // ---- trajectory recording: the post-step records to the caller's slot, in the launch that computed them (the next
            // launch's input stays the engine's own block, written below); like the planes they go to a destination nobody
            // re-reads soon, hence the planes' store mode
            if (STEP && a.rec_out) {
                uint4 *rdst = a.rec_out + (FUSED ? step * a.rec_step_stride : 0) + k0 * KP_recq(a);
#pragma unroll
                for (int q = 0; q < QN; ++q)
                    if (lane + q * kWave < nq && lane + q * kWave >= halo * KP_recq(a)) {
                        const uint32_t *sp = L.state + loff[q];
                        store16<SM>(reinterpret_cast<int8_t *>(rdst + lane + q * kWave), make_uint4(sp[0], sp[1], sp[2], sp[3]));
                    }
            }

        }   // step loop

        if (KP_trace(a) && stamp[4] == 0) stamp[4] = __builtin_amdgcn_s_memrealtime();
        // ---- state block LDS -> HBM
        if (STEP && !(KP_ablate(a) & 8)) {
            uint4 *dst = (((kHR || kXW) && a.state_out) ? a.state_out : a.state) + k0 * KP_recq(a);
#pragma unroll
            for (int q = 0; q < QN; ++q)
                if (lane + q * kWave < nq && lane + q * kWave >= halo * KP_recq(a)) {
                    const uint32_t *sp = L.state + loff[q];
                    // plain stores: the records are re-read by the next launch, and sending them write-through as well
                    // measured slower (12.9 vs 12.2 us at 65 536 kitchens)
                    dst[lane + q * kWave] = make_uint4(sp[0], sp[1], sp[2], sp[3]);
                }
        }
        wave_sync();                                 // next iteration overwrites the LDS state block
    }
    if (KP_trace(a)) {
        stamp[5] = __builtin_amdgcn_s_memrealtime();
        if (lane < 10) KP_trace(a)[static_cast<size_t>(blockIdx.x) * 16 + lane] = stamp[lane];
        if (lane == 10) KP_trace(a)[static_cast<size_t>(blockIdx.x) * 16 + 10] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID
    }
}


}  // namespace OC_NS
}  // namespace oc_dev
