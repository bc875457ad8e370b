// policy_ops.hip -- fused policy inference on the CDNA4 matrix cores, gfx950.
//
// One launch evaluates every (row range, network) tile of a rollout step: all conventions' actors and critics,
// both sides, every pairing segment -- the work the reference does as 2(2k+1) sequential CNN forwards of several
// launches each (train/XD/xd_player.py:190-207, train/partner_agents.py:97-111, rMAPPOPolicy.get_actions).
// Input rows are the engine's int8 observations exactly as oc_step wrote them ([rows][W][H][20], C fastest);
// outputs are sampled actions (int32, the layout oc_step reads), their log-probabilities, and critic values.
//
// Network = reference CNNBase + Linear head (train/MAPPO/utils/cnn.py:12-60; hidden_size 64).  Per 32 rows:
//   conv 3x3  : 6 x digits  v_mfma_i32_32x32x32_i8 per conv position.  Observations are small integers, so they
//               are fed as int8 without conversion; the f32 weights are split once, at pack time, into `digits`
//               base-256 fixed-point digits per output channel (policy_pack.h).  Integer accumulation is exact;
//               with 3 digits the weight quantisation (2^-23 of the channel's largest weight: the f32 spacing of
//               that weight) is of the class of the f32 rounding of the reference's own accumulation.
//   fc1/fc2/head : v_mfma_f32_32x32x16_f16 on two-way f16 splits of both operands (round to nearest: hi + lo carries 2^-24 of
//               the value, f32's own precision, as long as the low piece is a normal f16 number -- arranged with power-of-two
//               scales per layer, policy_pack.h), three piece products per f32 product (hi hi, hi lo, lo hi), f32 accumulate:
//               a 32x32x16 tile costs 3 x 32 cycles where rounds 2-3 (three bf16 pieces, six products) paid 6 x 32 and
//               v_mfma_f32_32x32x2_f32 needs 8 x 64.  The weights are split at pack time, the activations in the kernel: two
//               packed conversions and two v_fma_mix per register pair.
// Everything is evaluated transposed (feature index on the accumulator registers, row on the lanes): the output
// of one layer is then directly the B operand of the next, with no lane movement and no LDS round trip.
//
// Bound: MFMA issue + the VALU work between the layers (tools/probes/coexec.hip: they do not overlap on one SIMD).
// MFMA cycles per 32 rows on one SIMD (P conv positions): P*(6*digits*32 + 12*32) + 24*32 + 12*32; cramped_room
// (P = 6, 3 digits) 6912 cycles (10 368 with three bf16 pieces in round 3, 11 520 with four base-128 digits in round 2, 23 040 with f32
// MFMAs for the dense layers in round 1).  Bytes: rows are read once (W*H*20 B each).
//
// Shape of the kernel (DESIGN.md 9a has the measurements behind each choice):
//   * a workgroup = up to 8 wavefronts (two per SIMD; the kernel is held to 256 VGPRs and built with
//     -amdgpu-mfma-vgpr-form so MFMA results are VGPRs the VALU can read) working on rows of ONE network at a time;
//     which rows: oc_policy_plan_tiles balances one tile per CU, or the kernel splits a short table's 32-row blocks evenly;
//   * LDS: the network's conv digits and dense-layer weights (copied verbatim from the packed image by LDS-DMA; fc1
//     only when it fits, otherwise prefetched from L2 per position) + one 32-row staging area per wavefront, filled by
//     LDS-DMA one block ahead;
//   * per 32-row block a hand-pipelined loop over the conv positions (conv of p+1 and the operand conversion for p+1 in the
//     shadow of fc1 of p, fragments of p+2 in flight), then fc2, head and an epilogue in the lanes that hold the outputs.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

#include "overcooked_ops.h"
#include "overcooked_experiments.h"
#include "overcooked_debug.h"
#include "oc_internal.h"
#include "policy_pack.h"
#ifdef OC_DIAGNOSTICS
#include "oc_step_kernel.h"     // the generic env step body: folded into the policy launch (FOLD builds, diagnostics only)
#endif


namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

constexpr int kRowsPerBlock = 32;       // rows per wavefront pass = MFMA column count
#ifdef OC_DIAGNOSTICS
constexpr bool kFoldByDefault = false;   // oc_policy_rollout: two launches per step unless OC_POLICY_FOLD=1 asks for the env step inside the policy launch
#endif
constexpr int kMaxOut = 8;
#ifndef OC_POLICY_LATE_ROWS
#define OC_POLICY_LATE_ROWS 1
#endif
constexpr bool kLateRows = OC_POLICY_LATE_ROWS != 0;
constexpr int kCoopTailPositions = 3;  // cooperative blocks: the dense tail of a block in units of one conv position's time   // see the network change in oc_policy_kernel

constexpr int kWaveTileBlocks = 12;    // segments shorter than this (in 32-row blocks, on average) run as wave tiles (oc_policy_plan_tiles_multi)
constexpr int kMaxSlots = 8;           // observation tensors (grids) one launch can serve: tile flags carry the slot index

// One observation tensor with the networks that read it and the buffers they write (oc_policy_forward: one;
// oc_policy_forward_multi: one per layout of an engine group)
struct PolicySlot {
    const uint8_t *packed;              // [n_nets][L.total]
    oc_policy_layout L;                 // section offsets, identical for every network of the slot
    int n_nets;
    int RB;                             // bytes per row = W*H*20
    const int8_t *obs;                  // [n_rows][RB]
    long long n_rows;
    float *raw;                         // [n_rows][n_out] raw head outputs (tiles without OC_POLICY_TILE_ACT)
    int32_t *actions;                   // [n_rows]
    float *action_logp;                 // [n_rows]
    float *logp;                        // [n_rows][n_out] or NULL
    const int32_t *row_select;          // [n_rows] or NULL: tiles with a select code only write rows whose entry matches
    const int32_t *row_index;           // gather tiles: [lists][gather_len] row ids, class 0 first (oc_partition_rows), or NULL
    const int32_t *row_count;           // gather tiles: [lists][2] entries of either class
    int gather_len;                     // ids per list
    int gather_magic;                   // ceil(2^24 / 16-byte pieces per padded LDS row): piece index -> row of the block
    uint32_t seed;
    int pad_;
};

struct PolicyArgs {
    PolicySlot slot[kMaxSlots];
    int n_slots;
    const int4 *tiles;                  // {row0, nrows, net, flags | slot << 16}
    int n_tiles;
    int tile_per_wg;                    // K >= 1: workgroup i owns tiles [i K, (i + 1) K) (empty ones skipped); 0: the table's blocks are split evenly over the grid
    int desync;                         // experiment: the second wavefront of every SIMD starts this many x 64 cycles late
    int coop;                           // a stretch's last one or two blocks are shared by four (two) wavefronts, see "cooperative blocks" in oc_policy_kernel
    uint32_t counter;
    const uint32_t *counter_ptr;        // device counter added to `counter` (graph replay: a new draw per replay), or NULL
    int sample;                         // 0: argmax, 1: a draw from the distribution (inverse CDF of one counter-hash uniform per row)
    int lds_weight_bytes;               // LDS-resident part of the weight image, padded to whole KB (the copy runs in 1 KB instructions)
    int lds_wave_bytes;                 // one wavefront's row-staging area (the largest slot's)
    int coop1;                          // eight-wavefront builds: 1 = a stretch of ONE block is shared by four wavefronts too (0: only behind full rounds)
    int coop2;                          // eight-wavefront builds: group size for TWO leftover blocks (0: not shared, the default; 2: two pairs)
    int mono;                           // one-wavefront-per-SIMD builds: the position loop pipelined inside the wavefront (policy_mono_loop.h); 0: the plain loop (OC_POLICY_MONO=0)
    int lds_flag_offset;                // eight-wavefront plans: 64 bytes of hand-shake words behind the staging areas (cooperative blocks by flags), or 0
    unsigned long long *trace;          // diagnostics (oc_policy_debug_trace): [workgroup*8 + wave][32] stamps, or NULL
#ifdef OC_DIAGNOSTICS
    // FOLD builds (diagnostics library only): the env step of slot 0's engine runs inside this launch (see fold_* below)
    const oc_dev::StepArgs *fold_args;  // device copy of the engine's step arguments, fold geometry, without the per-step pointers ...
    const void *fold_actions;           // ... which are these: the action buffer this launch fills, and the step's outputs
    int8_t *fold_obs;
    int32_t *fold_reward, *fold_done;
    long long fold_kitchens;            // N
    int32_t *fold_count;                // [fold_groups] arrivals so far (zero between launches)
    const int32_t *fold_expected;       // [fold_groups] actor blocks that touch the group (fold_expected_kernel)
    int fold_groups;                    // ceil(N / kFoldKitchens)
    int fold_nt;                        // the env step's plane stores: 0 write-through (in place), 1 write-through non-temporal (fresh slot)
#endif
};

// (phase stamps exist in the diagnostics build only; in the product library the calls compile to nothing)
#if defined(OC_DIAGNOSTICS) || defined(OC_FORCE_STAMPS)
constexpr bool kPolicyTrace = true;
#else
constexpr bool kPolicyTrace = false;
#endif
__device__ __forceinline__ void stamp(const PolicyArgs &a, int wave, int lane, int slot) {
    if (kPolicyTrace && a.trace && lane == 0)
        a.trace[(static_cast<size_t>(blockIdx.x) * 8 + wave) * 32 + slot] = (slot == 14 || slot == 15) ? clock64() : wall_clock64();
}

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef __attribute__((address_space(4))) const uint32_t *kptr_t;      // constant address space: scalar loads

// The launch arguments are 1.7 KB of kernarg memory that the prologue reads a few dwords at a time, each read depending on the
// last (tile_per_wg -> the table pointer -> the tile -> its slot's row count ...): five scalar-cache misses one after the other,
// ~0.5 us each at the start of a launch.  Touch every 64-byte line the launch will use at once, so that the chain runs on hits.
template <bool ALL_SLOTS>
__device__ __forceinline__ void warm_kernarg() {
    const kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int kSlot = static_cast<int>(sizeof(PolicySlot)), kAll = static_cast<int>(sizeof(PolicyArgs));
    uint32_t x = 0;
#pragma unroll
    for (int off = 0; off < kAll; off += 64)
        if (ALL_SLOTS || off < kSlot + 64 || off + 64 > kMaxSlots * kSlot) x |= ka[off / 4];
    asm volatile("" ::"s"(x));
}
// The tile table does not change during a launch and every index into it is wave-uniform: read it with scalar loads.
__device__ __forceinline__ int4 load_tile(const int4 *tiles, int i) {
    const kptr_t p = (kptr_t)(tiles + i);
    return make_int4(static_cast<int>(p[0]), static_cast<int>(p[1]), static_cast<int>(p[2]), static_cast<int>(p[3]));
}

// Kernel-argument preload (policy_ops.hip is built with -amdgpu-kernarg-preload-count=4): the table pointer and the tiles-per-workgroup
// count arrive in SGPRs with the wavefront, in front of the 1.7 KB argument struct, so a workgroup's own table entry is requested in the
// kernel's first instructions, beside the struct's lines (warm_kernarg) instead of one scalar-cache miss behind them; the stretch search
// then reads it from the scalar cache.
__device__ __forceinline__ int prefetch_own_tile(const int4 *pre_tiles, int pre_tile_per_wg) {
    const kptr_t p = (kptr_t)(pre_tiles + (pre_tile_per_wg > 0 ? static_cast<int>(blockIdx.x) * pre_tile_per_wg : 0));
    return static_cast<int>(p[0]);                                   // (requested here; keep_prefetched() behind warm_kernarg is its only use)
}
__device__ __forceinline__ void keep_prefetched(int v) { asm volatile("" ::"s"(v)); }

// Hand-shake words in LDS between SOME wavefronts of a workgroup (s_barrier takes all of them): a 4-byte LDS word written by one
// wavefront and polled by others.  LDS requests of a CU are served in order, and an LDS-DMA write has happened when the issuing
// wavefront's vmcnt has counted it, so whatever the writer finished (and waited for) before the store is there for whoever sees the word.
__device__ __forceinline__ void lds_flag_store(uint8_t *lds, int off, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(reinterpret_cast<int *>(lds + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_add(uint8_t *lds, int off, int v, int lane) {   // (ONE lane: an LDS atomic adds once per active lane)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<int *>(lds + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_wait(uint8_t *lds, int off, int want) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(reinterpret_cast<int *>(lds + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != want)
        __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The value the lane 32 places away holds (__shfl_xor(x, 32)): v_permlane32_swap exchanges the upper half of one register with the
// lower half of another -- a vector instruction and a select where ds_bpermute_b32 is an LDS round trip (~100 cycles on the epilogue's
// dependent chain: maximum -> sum -> chosen action, tools/probes/permlane32_swap.hip).  `upper`: lane >= 32.
__device__ __forceinline__ unsigned xor32_bits(unsigned u, bool upper) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u r = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // r.x = {lower, lower}, r.y = {upper, upper}
    return upper ? r.x : r.y;
}
__device__ __forceinline__ float xor32(float x, bool upper) { return __builtin_bit_cast(float, xor32_bits(__builtin_bit_cast(unsigned, x), upper)); }
__device__ __forceinline__ int xor32(int x, bool upper) { return static_cast<int>(xor32_bits(static_cast<unsigned>(x), upper)); }

// f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>), in order: compile-time indices into register arrays
template <int... Is, class F>
__device__ __forceinline__ void oc_static_for_impl(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>()), ...); }
template <int N, class F>
__device__ __forceinline__ void oc_static_for(F &&f) { oc_static_for_impl(std::make_integer_sequence<int, N>(), static_cast<F &&>(f)); }

__device__ __forceinline__ void wave_lds_sync() {
    // LDS traffic of one wavefront executes in program order; this only stops the compiler from moving LDS
    // accesses across the hand-over between "stage rows" and "read fragments"
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Copy 32 rows (one flat, contiguous block of the observation tensor) into this wavefront's staging area with
// LDS-DMA loads: no VGPRs, so the copy for the NEXT block can be in flight under the dense layers of this one.
// Pieces past the end of the tensor are skipped (their rows are never stored); the one piece that straddles the
// end goes dword by dword.
// Returns the number of vector-memory instructions issued (for a counted vmcnt wait), or -1 on the ragged path.
__device__ __forceinline__ int stage_rows(const PolicySlot &a, uint8_t *ldsw, long long brow, int lane) {
    const long long obs_bytes = a.n_rows * a.RB, base = brow * a.RB;
    const int n16 = 2 * a.RB;                                         // 16-byte pieces in 32 rows
    // Fast path: whole 1 KB instructions only.  The staging area is padded to a multiple of 1 KB (lds_wave_bytes), so the last
    // instruction may run past the block's 32 rows into the following rows of the tensor (read, never used).  Eight
    // instructions share one LDS base (M0), told apart by the signed 13-bit immediate offset that is added to both addresses:
    // every change of M0 holds the wavefront ~0.12 us behind the copies in flight (one M0 per instruction measured 1.5 us for
    // the 13 of a cramped_room block), and hipcc re-writes M0 in every basic block -- hence straight-line groups.
    const int n_kb = (n16 + 63) >> 6;
    if (base + 1024ll * n_kb <= obs_bytes) {                          // (wave-uniform)
        const int8_t *src = a.obs + base + 16 * lane;
#define OC_DMA(J) __builtin_amdgcn_global_load_lds((gptr_t)s4, (lptr_t)d4, 16, ((J) - 4) * 1024, 0);
        int it = 0;
        for (; it + 8 <= n_kb; it += 8) {
            const int8_t *s4 = src + (it + 4) * 1024;
            uint8_t *d4 = ldsw + (it + 4) * 1024;
            OC_DMA(0) OC_DMA(1) OC_DMA(2) OC_DMA(3) OC_DMA(4) OC_DMA(5) OC_DMA(6) OC_DMA(7)
        }
        const int8_t *s4 = src + (it + 4) * 1024;
        uint8_t *d4 = ldsw + (it + 4) * 1024;
        switch (n_kb - it) {                                          // the remaining 0..7, one M0 for all of them
        case 7: OC_DMA(0) OC_DMA(1) OC_DMA(2) OC_DMA(3) OC_DMA(4) OC_DMA(5) OC_DMA(6) break;
        case 6: OC_DMA(0) OC_DMA(1) OC_DMA(2) OC_DMA(3) OC_DMA(4) OC_DMA(5) break;
        case 5: OC_DMA(0) OC_DMA(1) OC_DMA(2) OC_DMA(3) OC_DMA(4) break;
        case 4: OC_DMA(0) OC_DMA(1) OC_DMA(2) OC_DMA(3) break;
        case 3: OC_DMA(0) OC_DMA(1) OC_DMA(2) break;
        case 2: OC_DMA(0) OC_DMA(1) break;
        case 1: OC_DMA(0) break;
        default: break;
        }
#undef OC_DMA
        return n_kb;
    }
    for (int it = 0; it * 64 < n16; ++it) {
        const int i = it * 64 + lane;
        const long long off = base + 16ll * i;
        if (i < n16 && off + 16 <= obs_bytes) {
            __builtin_amdgcn_global_load_lds((gptr_t)(a.obs + off), (lptr_t)(ldsw + it * 1024), 16, 0, 0);
        } else if (i < n16 && off < obs_bytes) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (off + 4 * q + 4 <= obs_bytes)
                    reinterpret_cast<uint32_t *>(ldsw + 16 * i)[q] = *reinterpret_cast<const uint32_t *>(a.obs + off + 4 * q);
        }
    }
    return -1;
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
#define OC_W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
    OC_W(1) OC_W(2) OC_W(3) OC_W(4) OC_W(5) OC_W(6) OC_W(7) OC_W(8) OC_W(9) OC_W(10) OC_W(11) OC_W(12) OC_W(13) OC_W(14) OC_W(15) OC_W(16)
    OC_W(17) OC_W(18) OC_W(19) OC_W(20) OC_W(21) OC_W(22) OC_W(23) OC_W(24) OC_W(25) OC_W(26) OC_W(27) OC_W(28) OC_W(29) OC_W(30) OC_W(31) OC_W(32)
#undef OC_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// The same for GATHERED rows (OC_POLICY_TILE_GATHER: the rows a tile evaluates are listed in an index array -- mixed play gives
// every row to one of two policies, and each policy's tile reads only its own rows).  LDS-DMA writes lane l's 16 bytes at
// base + 16 l whatever the lane's global address is, so the staging area is filled flat, 1 KB per instruction, exactly as above,
// and only the global side is indirect: piece f of the block belongs to row f / ppr of the block (ppr = 16-byte pieces per
// row, rows padded to whole pieces in LDS) and comes from that row's address in the observation tensor.  A row of a grid with
// W*H % 4 != 0 is not a whole number of pieces: its last piece reads up to 12 bytes of the following row (global_load_lds
// takes dword-aligned addresses on both sides, tools/probes/lds_dma_align.hip); for the tensor's very last row it reads the row's
// last 16 bytes instead and the consumer repairs those few bytes at its landing wait.  Eight instructions share one LDS base
// (M0) through the immediate offset, which is added to BOTH addresses -- hence subtracted from the global one first.
// Returns the row id held by lane & 31 (rows past the tile's end repeat its last row: read, never stored); n_instr: LDS-DMA instructions issued.
__device__ __forceinline__ int stage_rows_gather(const PolicySlot &a, uint8_t *ldsw, const int32_t *gidx, int first, int nrows, int lane, int &n_instr) {
    const int ppr = (a.RB + 15) >> 4, total = 32 * ppr, n_kb = (total + 63) >> 6;
    n_instr = n_kb;
    const int my = gidx[min(first + (lane & 31), nrows - 1)];
    const bool ragged = (a.RB & 15) != 0;
    const int last_row = static_cast<int>(a.n_rows - 1);
    auto src_of = [&](int it) -> const int8_t * {
        const int f = min(it * 64 + lane, total - 1);                 // (lanes past the block re-read its last piece into the slack)
        const int row = static_cast<int>((static_cast<uint32_t>(f) * static_cast<uint32_t>(a.gather_magic)) >> 24);
        const int piece = f - row * ppr;
        const int id = __builtin_amdgcn_ds_bpermute(row << 2, my);
        int off = piece * 16;
        if (ragged) off = (id == last_row && piece == ppr - 1) ? a.RB - 16 : off;
        return a.obs + static_cast<long long>(id) * a.RB + off;
    };
#define OC_DMAG(J) { const int8_t *sj = src_of(it + (J)) - ((J) - 4) * 1024; __builtin_amdgcn_global_load_lds((gptr_t)sj, (lptr_t)d4, 16, ((J) - 4) * 1024, 0); }
    int it = 0;
    for (; it + 8 <= n_kb; it += 8) {
        uint8_t *d4 = ldsw + (it + 4) * 1024;
        OC_DMAG(0) OC_DMAG(1) OC_DMAG(2) OC_DMAG(3) OC_DMAG(4) OC_DMAG(5) OC_DMAG(6) OC_DMAG(7)
    }
    uint8_t *d4 = ldsw + (it + 4) * 1024;
    switch (n_kb - it) {
    case 7: OC_DMAG(0) OC_DMAG(1) OC_DMAG(2) OC_DMAG(3) OC_DMAG(4) OC_DMAG(5) OC_DMAG(6) break;
    case 6: OC_DMAG(0) OC_DMAG(1) OC_DMAG(2) OC_DMAG(3) OC_DMAG(4) OC_DMAG(5) break;
    case 5: OC_DMAG(0) OC_DMAG(1) OC_DMAG(2) OC_DMAG(3) OC_DMAG(4) break;
    case 4: OC_DMAG(0) OC_DMAG(1) OC_DMAG(2) OC_DMAG(3) break;
    case 3: OC_DMAG(0) OC_DMAG(1) OC_DMAG(2) break;
    case 2: OC_DMAG(0) OC_DMAG(1) break;
    case 1: OC_DMAG(0) break;
    default: break;
    }
#undef OC_DMAG
    return my;
}

// B operands of the conv at one position: for each kernel column kx the 60 contiguous bytes of tiles
// (x'+kx, y'..y'+2) plus 4 bytes that meet zero weights, as 2 x 16 bytes per lane half.
__device__ __forceinline__ void read_bfrags(const uint8_t *rowp, int pos_off, int colstep, v4i (&bfr)[6]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(rowp + pos_off + kx * colstep + 32 * s);
            v4i b;
            b.x = static_cast<int>(q[0]); b.y = static_cast<int>(q[1]);
            b.z = static_cast<int>(q[2]); b.w = static_cast<int>(q[3]);
            bfr[kx * 2 + s] = b;
        }
}

// One dense-layer tile step: D += A (weight pieces, 8 f16 per lane) x B (activation pieces, 8 f16 per lane)
__device__ __forceinline__ v16f mfma_f16(const v4i a, const v4i b, const v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
}
// The three piece products kept of (w_hi + w_lo) * (x_hi + x_lo): weight piece, activation piece (lo * lo is below 2^-24 of the product)
constexpr int kPieces = OC_POLICY_PIECES, kTerms = 3;
__device__ constexpr int kTermW[kTerms] = {0, 0, 1};
__device__ constexpr int kTermX[kTerms] = {0, 1, 0};

// An activation as a dense-layer operand: relu, and the clamp to f16's largest finite number, in ONE instruction -- v_med3_f32 through
// the builtin.  NOT inline asm: a VALU instruction that reads a register an MFMA has just written needs up to 19 wait states behind it
// on gfx950, the compiler's hazard recogniser inserts them for instructions it can see and does not look inside asm -- as
// `asm("v_max_f32 ...")` (rounds 2-3) the first relu behind fc2's last MFMA read a stale accumulator whenever nothing else happened to
// be scheduled in between: the wave-tile build of the product library (no LDS read there, and no phase stamps since round 4) returned
// values off by up to 0.1 until this was found.  (fmaxf would first quiet a possible signalling NaN with a second v_max.)
__device__ __forceinline__ float relu_clamp(float x) {
    return __builtin_amdgcn_fmed3f(x, 0.0f, OC_POLICY_F16_MAX);
}

// Two f32 activations (already relu'd, clamped and in the operand scale) -> one packed dword per f16 piece, x = hi + lo to 2^-24 of x:
// hi = f16(x) and lo = f16(x - hi), both round to nearest even (v_cvt_pk_f16_f32); the remainder is exact (v_fma_mix_f32 reads the f16
// half as an f32 operand: fma(hi, -1, x), one instruction per value).  4 VALU per pair where the three-way bf16 split cost 7.
__device__ __forceinline__ void split_pair(float x0, float x1, int &hi, int &lo) {
    const v2f xv = {x0, x1};
    const v2h h = __builtin_convertvector(xv, v2h);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    const v2f rv = {r0, r1};
    const v2h l = __builtin_convertvector(rv, v2h);
    hi = __builtin_bit_cast(int, h);
    lo = __builtin_bit_cast(int, l);
}

// fragment f (0..7) of position p of the fc1 section: voffset = lane * 16 (+ immediate), soffset scalar
__device__ __forceinline__ v4i w1_from_l2(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane16, int p, int f) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(lane16) + (f & 3) * 1024,
                                                         p * OC_POLICY_FC1_POS_BYTES + (f >> 2) * 4096, 0);
    return __builtin_bit_cast(v4i, v);
}

// One contiguous part of a network's image -> LDS, in whole 1 KB LDS-DMA instructions (lane l of instruction c moves bytes
// [c * 1024 + 16 l, + 16)): every wavefront takes a contiguous share of the instructions and issues them eight per LDS base (M0),
// told apart by the immediate offset like stage_rows -- a change of M0 holds the wavefront behind the copies in flight, and with
// one base per instruction the seven instructions of a wavefront went out one round trip after the other (3 us of a 6 us
// prologue).  The part's last instruction may run past its end: the first `last_valid` lanes carry the part, the others the
// first 16-byte pieces of `hdr` (64: nothing is replaced and the instruction reads on into whatever follows the part).
__device__ __forceinline__ void copy_to_lds(const uint8_t *src, uint8_t *dst, int nk, int last_valid, const uint8_t *hdr, int wave, int nwaves, int lane) {
    const int per = (nk + nwaves - 1) / nwaves, k1 = min((wave + 1) * per, nk);
    for (int g = wave * per; g < k1; g += 8) {
        const uint8_t *sv = src + (g + 4) * 1024 + lane * 16;         // instruction J of the group: sv + (J - 4) * 1024
        uint8_t *d4 = dst + (g + 4) * 1024;
        const uint8_t *sl = sv;                                       // the same for the part's last instruction, should it need the header
        if (last_valid < 64 && nk - 1 < g + 8)
            sl = (lane < last_valid ? sv : hdr + 16 * min(lane - last_valid, OC_POLICY_HEADER_BYTES / 16 - 1) - (nk - 1 - g - 4) * 1024);
#define OC_DMAW(J) { const uint8_t *sj = g + (J) == nk - 1 ? sl : sv; __builtin_amdgcn_global_load_lds((gptr_t)sj, (lptr_t)d4, 16, ((J) - 4) * 1024, 0); }
        switch (min(k1 - g, 8)) {                                     // (straight-line groups: hipcc re-writes M0 in every basic block)
        case 8: OC_DMAW(0) OC_DMAW(1) OC_DMAW(2) OC_DMAW(3) OC_DMAW(4) OC_DMAW(5) OC_DMAW(6) OC_DMAW(7) break;
        case 7: OC_DMAW(0) OC_DMAW(1) OC_DMAW(2) OC_DMAW(3) OC_DMAW(4) OC_DMAW(5) OC_DMAW(6) break;
        case 6: OC_DMAW(0) OC_DMAW(1) OC_DMAW(2) OC_DMAW(3) OC_DMAW(4) OC_DMAW(5) break;
        case 5: OC_DMAW(0) OC_DMAW(1) OC_DMAW(2) OC_DMAW(3) OC_DMAW(4) break;
        case 4: OC_DMAW(0) OC_DMAW(1) OC_DMAW(2) OC_DMAW(3) break;
        case 3: OC_DMAW(0) OC_DMAW(1) OC_DMAW(2) break;
        case 2: OC_DMAW(0) OC_DMAW(1) break;
        default: OC_DMAW(0) break;
        }
#undef OC_DMAW
    }
}

// ---- work distribution, shared by the kernels below.  The tile table lists (row range, network, flags); its 32-row
// blocks, in table order, are split EVENLY over the workgroups of the launch (one per CU), so a workgroup owns a contiguous
// stretch of blocks [wg0, wg1) that may run through several tiles (weights are reloaded only when the network changes along
// it); a planned table -- about one balanced tile per CU, oc_policy_plan_tiles -- maps tile i to workgroup i instead.
// Everything is wave-uniform.  Returns false when the workgroup has no block.
// rows a gather tile {list, class, net, flags} evaluates: what oc_partition_rows counted for that class of that list
__device__ __forceinline__ int policy_gather_rows(const PolicySlot &S, const int4 t) {
    if (!S.row_index || !S.row_count || t.x < 0) return 0;
    return min(max(S.row_count[2 * t.x + (t.y & 1)], 0), S.gather_len);
}
template <bool GATHER>
__device__ __forceinline__ int policy_tile_blocks(const PolicySlot &S, const int4 t) {
    if (t.w & OC_POLICY_TILE_GATHER) return GATHER ? (policy_gather_rows(S, t) + kRowsPerBlock - 1) / kRowsPerBlock : 0;
    long long nr = t.y;
    if (t.x < 0 || t.x >= S.n_rows || nr <= 0) return 0;            // a bad tile contributes nothing
    if (t.x + nr > S.n_rows) nr = S.n_rows - t.x;
    return static_cast<int>((nr + kRowsPerBlock - 1) / kRowsPerBlock);
}

template <bool GATHER>
__device__ __forceinline__ bool policy_find_stretch(const PolicyArgs &a, int lane, long long &wg0, long long &wg1, int &ti, long long &tstart) {
    // Every wavefront scans the table itself, 64 entries per pass (one coalesced load + a wave prefix sum; a scalar
    // loop costs one dependent L2 round trip per entry), first for the total, then for the entry holding its first block.
    auto scan64 = [&](int base_tile, int &nb, int &incl) {
        const int i = base_tile + lane;
        int4 t = make_int4(0, 0, 0, 0);
        if (i < a.n_tiles) t = a.tiles[i];
        nb = i < a.n_tiles ? policy_tile_blocks<GATHER>(a.slot[0], t) : 0;             // (launches over several slots are always tile-per-workgroup)
        incl = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
    };
    long long total = 0;
    int nb0 = 0, incl0 = 0;                                           // the first 64 entries' counts, kept for the second pass: one round trip less
    if (!a.tile_per_wg)
        for (int c = 0; c < a.n_tiles; c += 64) {
            int nb, incl;
            scan64(c, nb, incl);
            if (c == 0) { nb0 = nb; incl0 = incl; }
            total += __builtin_amdgcn_readlane(incl, 63);
        }
    wg0 = a.tile_per_wg ? 0 : total * blockIdx.x / gridDim.x;
    if (a.tile_per_wg) {                                              // (several entries per workgroup as ONE vector load: measured, no faster)
        wg1 = 0;
        for (int j = 0; j < a.tile_per_wg; ++j) {                    // (wave-uniform)
            const int4 t = load_tile(a.tiles, blockIdx.x * a.tile_per_wg + j);
            wg1 += policy_tile_blocks<GATHER>(a.slot[a.n_slots > 1 ? __builtin_amdgcn_readfirstlane((t.w >> 16) & (kMaxSlots - 1)) : 0], t);
        }
    } else {
        wg1 = total * (blockIdx.x + 1) / gridDim.x;
    }
    ti = a.tile_per_wg ? static_cast<int>(blockIdx.x) * a.tile_per_wg : 0;
    tstart = 0;                                                       // first block of tile ti
    if (wg0 >= wg1) return false;
    if (!a.tile_per_wg)
        for (int c = 0; c < a.n_tiles; c += 64) {
            int nb = nb0, incl = incl0;
            if (c) scan64(c, nb, incl);
            const long long target = wg0 - tstart;
            const unsigned long long hit = __ballot(static_cast<long long>(incl - nb) <= target && target < incl);
            if (hit) {
                const int src = __ffsll(static_cast<long long>(hit)) - 1;
                ti = c + src;
                tstart += __builtin_amdgcn_readfirstlane(__shfl(incl - nb, src));
                break;
            }
            tstart += __builtin_amdgcn_readlane(incl, 63);
        }
    ti = __builtin_amdgcn_readfirstlane(ti);
    return true;
}

#ifdef OC_DIAGNOSTICS
// ---- the env step inside the policy launch ("one launch per training step"; diagnostics build only: measured 2x slower than two
// launches) -------------------------------------------------
// The reference's loop is policy -> env -> policy (train/XD/xd_player.py:178-230); as two launches per step the env launch is a
// kernel of its own that the matrix pipe waits out.  FOLD builds run it inside the policy launch, overlapped with other
// wavefronts' MFMA work: kitchens are cut into fold groups of kFoldKitchens; every actor block, once its actions are stored and
// fenced, adds one to the arrival counter of each group it touches, and the wavefront whose add completes a group (the counter
// reaches the number of actor blocks touching it, counted once per rollout by fold_expected_kernel) steps that group with the
// generic step body -- transition, encode, plane / state / reward / done stores -- using its own row-staging area as the body's
// LDS block, at the next point where that area is free.  Nobody waits for anybody: the last arriver does the work.
// Groups touched by rows [r0, r1] of the [2][N] row space (a block never spans more than 32 rows, so at most four).
__device__ __forceinline__ int fold_block_groups(long long r0, long long r1, long long N, int (&g)[4]) {
    int n = 0;
    auto span = [&](long long ka, long long kb) {
        for (int q = static_cast<int>(ka / kFoldKitchens); q <= static_cast<int>(kb / kFoldKitchens) && n < 4; ++q) {
            bool seen = false;
            for (int i = 0; i < n; ++i) seen |= g[i] == q;
            if (!seen) g[n++] = q;
        }
    };
    const long long s0 = r0 / N, s1 = r1 / N;
    if (s0 == s1) span(r0 - s0 * N, r1 - s0 * N);
    else { span(r0 - s0 * N, N - 1); span(0, r1 - s1 * N); }
    return n;
}

// expected[g] = number of (actor block, group) incidences of the tile table: one thread per tile walks its blocks
__global__ void fold_expected_kernel(const int4 *tiles, int n_tiles, long long n_rows, long long N, int32_t *expected) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    const int4 t = tiles[i];
    if (!(t.w & OC_POLICY_TILE_ACT) || t.x < 0 || t.x >= n_rows || t.y <= 0) return;
    const long long end = t.x + t.y > n_rows ? n_rows : static_cast<long long>(t.x) + t.y;
    for (long long r = t.x; r < end; r += kRowsPerBlock) {
        int g[4];
        const int n = fold_block_groups(r, (r + kRowsPerBlock < end ? r + kRowsPerBlock : end) - 1, N, g);
        for (int j = 0; j < n; ++j) atomicAdd(expected + g[j], 1);
    }
}

// One fold group stepped by the calling wavefront: sub-groups g * kSub .. of the generic step body, one after the other, with
// `lds` (the wavefront's row-staging area) as the body's LDS block.  A real call, not inlined: the body then gets its own register
// allocation instead of being woven -- four copies of it -- into a kernel that sits at its 256-register limit (inlined: 2.6 KB of
// scratch per lane, and a launch three times slower than two launches).
struct FoldStep {                       // what changes from step to step: by value, in registers
    const void *actions;
    int8_t *obs;
    int32_t *reward, *done;
};
// Everything but `lane` is wave-uniform, and says so (readfirstlane; the template is read through the constant address space):
// arguments of a real call arrive in vector registers, and a step body whose "scalars" live in VGPRs -- or, as a struct copy, in
// scratch -- ran twenty times slower than the same body as a kernel of its own.
template <typename T> __device__ __forceinline__ T *fold_uniform(T *p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return reinterpret_cast<T *>((static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __attribute__((noinline)) void fold_env_group(const oc_dev::StepArgs *tmpl, FoldStep st, int g, int nt, int lds_dword_offset, int lane, int32_t *count) {
    oc_dev::StepArgs ea;
    {
        typedef __attribute__((address_space(4))) const uint32_t *const_words_t;       // scalar loads
        const_words_t src = (const_words_t)fold_uniform(reinterpret_cast<const uint32_t *>(tmpl));
        uint32_t *dst = reinterpret_cast<uint32_t *>(&ea);
        static_assert(sizeof(oc_dev::StepArgs) % 4 == 0, "StepArgs is copied word by word");
#pragma unroll
        for (int i = 0; i < static_cast<int>(sizeof(oc_dev::StepArgs) / 4); ++i) dst[i] = src[i];
    }
    ea.actions = fold_uniform(st.actions); ea.obs = fold_uniform(st.obs); ea.reward = fold_uniform(st.reward); ea.done = fold_uniform(st.done);
    g = __builtin_amdgcn_readfirstlane(g);
    nt = __builtin_amdgcn_readfirstlane(nt);
    const int lds = __builtin_amdgcn_readfirstlane(lds_dword_offset);    // (an offset into the dynamic LDS, not a pointer: see step_body)
    count = fold_uniform(count);
    constexpr int kSub = kFoldKitchens / kFoldKpw;
    ea.n_groups = min(ea.n_groups, (g + 1) * kSub);
    if (nt & 1) oc_dev::generic::step_body<true, false, true, kFoldQN, oc_dev::SM_SC1_NT, false, false, true>(ea, g * kSub, 1, lds, lane);
    else oc_dev::generic::step_body<true, false, true, kFoldQN, oc_dev::SM_SC1, false, false, true>(ea, g * kSub, 1, lds, lane);
    if (lane == 0) count[g] = 0;                                      // every contributor has arrived: ready for the next launch
}
#endif  // OC_DIAGNOSTICS

// MULTI: the tile table runs over several slots (grids); everything that depends on the grid is then re-derived per tile.
// Such launches keep the conv digits in registers and fc1 in L2 (the LDS-resident tail -- biases, fc2, head -- has the same
// layout for every grid), and map workgroup i to tile i.
// TAIL_LDS = false ("wave tiles"): nothing of the network lives in LDS -- biases, fc2 and head fragments are read from L2 per
// block like fc1 -- so a workgroup is ONE wavefront with its own network, four independent ones per CU.  For tables of short
// segments (config 5: five 32-row blocks per segment and network), where a four-wavefront workgroup that must stay on one
// network leaves 3 of 8 wavefront slots empty.
// GATHER: the build that also serves OC_POLICY_TILE_GATHER tiles (launches whose slot carries index lists).  A build of its own:
// compiled into the one kernel the few extra live values cost the plain launches 7 % (48.1 -> 51.7 us for 131 070 evaluations:
// the kernel sits at the 256-register limit of two wavefronts per SIMD).
template <int NDIG, bool FC1_LDS, bool CONV_LDS, bool MULTI = false, bool TAIL_LDS = true, bool GATHER = false, bool FOLD = false>
__global__ __launch_bounds__(CONV_LDS ? 512 : 256) void oc_policy_kernel(const int4 *pre_tiles, int pre_tile_per_wg, const PolicyArgs a) {
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds_raw);
    // (readfirstlane: hipcc cannot prove threadIdx.x >> 6 wave-uniform; without it every block index, bound and LDS-DMA
    // base derived from it is computed per lane, under exec masks)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    uint8_t *ldsw = lds + a.lds_weight_bytes + static_cast<size_t>(wave) * a.lds_wave_bytes;

    // ---- work distribution.  The tile table lists (row range, network, flags); its 32-row blocks, in table order,
    // are split EVENLY over the workgroups of the launch (one per CU), so a workgroup owns a contiguous stretch of
    // blocks that may run through several tiles.  Weights are (re)loaded only when the network changes along the
    // stretch.  Everything here is wave-uniform (scalar loads of the table).
    long long wg0, wg1, tstart;
    int ti;
    const int own_tile_word = prefetch_own_tile(pre_tiles, pre_tile_per_wg);
    warm_kernarg<false>();                                           // (every slot's lines, for launches over several grids: measured slower -- the misses do not all overlap)
    keep_prefetched(own_tile_word);
    stamp(a, wave, lane, 9);
    if (!policy_find_stretch<GATHER>(a, lane, wg0, wg1, ti, tstart)) {
        if (!FOLD) return;
        wg0 = wg1 = 0;                                                // no policy work: still takes its share of the untouched fold groups
    }

    // grid-dependent values: of slot 0, or (MULTI) re-derived at every tile
    int cur_slot = 0;
    int RB = a.slot[0].RB, P = a.slot[0].L.P, PH = a.slot[0].L.PH, H = a.slot[0].L.H;
    const uint32_t counter = a.counter + (a.counter_ptr ? *a.counter_ptr : 0u);
    // image offset of the LDS tail (MULTI: off_b1 differs per slot, but everything behind it has the same layout)
    int wstart = static_cast<int>(FC1_LDS ? a.slot[0].L.off_fc1 : a.slot[0].L.off_b1);
    const oc_policy_layout &L0 = a.slot[0].L;
    const int conv_bytes = CONV_LDS ? static_cast<int>(L0.off_fc1 - L0.off_conv) : 0;
    uint8_t *lds_tail = lds + ((conv_bytes + 1023) & ~1023);                          // LDS address of image offset wstart (both parts start on a KB)
    // conv weight digits: [kx*2+s][digit] x 16 bytes per lane, read from LDS per use or (when the LDS is needed for
    // large rows) held in registers; conv scale / bias per accumulator register likewise
    v4i wA[CONV_LDS ? 1 : 6][NDIG];
    // conv bias in units of the channel's fixed-point scale (policy_pack.h): the lowest digit's accumulator starts from it
    v16i cbeta;
    const v4i *wcl = reinterpret_cast<const v4i *>(lds) + lane;
    const v16i *cbeta_lds = reinterpret_cast<const v16i *>(lds + (L0.off_conv_sb - L0.off_conv) + 128 + h * 64);
    // dense-layer weight fragments: 16 bytes (8 bf16) per lane and MFMA, fragment f of a section at [f * 64 + lane]
    // fc1 section of the network in global memory, as a buffer resource: scalar descriptor + scalar position offset + one
    // 32-bit lane offset per load (as plain pointers hipcc keeps twelve 64-bit vector addresses alive through the loop)
    __amdgpu_buffer_rsrc_t w1g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.slot[0].packed), 0, 0, 0x00020000);
    const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
    const v4i *w1l = reinterpret_cast<const v4i *>(lds_tail + (L0.off_fc1 - wstart)) + lane;
    const float *b1 = TAIL_LDS ? reinterpret_cast<const float *>(lds_tail + (L0.off_b1 - wstart)) + h * 16 : nullptr;
    const v4i *w2 = TAIL_LDS ? reinterpret_cast<const v4i *>(lds_tail + (L0.off_fc2 - wstart)) + lane : nullptr;
    const float *b2 = TAIL_LDS ? reinterpret_cast<const float *>(lds_tail + (L0.off_b2 - wstart)) + h * 16 : nullptr;
    // head: only the outputs a < 8 are stored, [piece][k-group][8 h + a]; the other rows of the A operand are zero registers
    const bool head_lane = col < 8;
    const v4i *wh = TAIL_LDS ? reinterpret_cast<const v4i *>(lds_tail + (L0.off_head - wstart)) + (8 * h + (col & 7)) : nullptr;
    const float *bh = TAIL_LDS ? reinterpret_cast<const float *>(lds_tail + (L0.off_bh - wstart)) + h * 4 : nullptr;
    const uint8_t *rowp = ldsw + col * RB + 16 * h;
    int colstep = H * OC_POLICY_C;
    int loaded_net = -1, loaded_slot = 0, n_out = 1;
    // the loaded network's descale factors (powers of two, policy_pack.h): fc1 accumulators -> fc2 operand, fc2 -> head operand, head -> outputs
    float ds1 = 1.f, ds2 = 1.f, ds3 = 1.f;
    bool trace_done = false, trace2_done = false;
    // The position loop pipelined inside the wavefront (policy_mono_loop.h), for the builds that run one wavefront per SIMD: built, bit-identical,
    // 5-7 % faster on large 9x5 launches -- and compiled into the DIAGNOSTICS library only (or with -DOC_POLICY_MONO_LOOP): at three conv
    // digits its two accumulator sets push the kernel past 512 registers, the spills land in scratch, and their reloads behind every block
    // cost config 4's per-GPU share (one block per wavefront) 7 us of 39 per step whichever loop runs.
#if defined(OC_DIAGNOSTICS) || defined(OC_POLICY_MONO_LOOP)
    constexpr bool kMono = !CONV_LDS && !FOLD && !GATHER;             // (the gather build with four digits: hipcc 7.2 crashes in 'AMDGPU Rewrite AGPR-Copy-MFMA')
#else
    constexpr bool kMono = false;
#endif
    stamp(a, wave, lane, 0); stamp(a, wave, lane, 14);

    // conv result -> fc1 operand: relu(2^-shift x f32(sum of digit accumulators)), two accumulator registers at a time.  The channel's
    // bias is already in the sum (the lowest digit's accumulator starts from it) and its scale is folded into the fc1 weights; the
    // power of two brings the integer sums (up to 2^31) into f16's range for the split (oc_policy_conv_shift: exact, and the same for
    // every channel, so it is a constant of the packed multiplies here, not a per-register value).  The top digit pair is combined in
    // integer arithmetic (acc_0 * B + acc_1: exact, |.| < 2^31), so three digits cost 1 shift-add + 2 conversions + two halves of a
    // packed multiply / fma per register: a wave's VALU work is not hidden under its MFMAs.
    auto combine_pair = [&](const v16i (&acc)[NDIG], int r, float &x0, float &x1) {
        // (plain f32 instructions, one per value: a packed v_pk_mul_f32 / v_pk_fma_f32 beside MFMAs costs the SIMD ~22 cycles more than
        // the two instructions it replaces -- MI355X guide, "price of one filler beside MFMAs"; the rounding is the same either way)
        constexpr int kB = OC_POLICY_DIGIT_BASE;
        const float sc = oc_pow2f(-oc_policy_conv_shift(NDIG));
        float t[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (NDIG == 1) {
                t[e] = static_cast<float>(acc[0][r + e]) * sc;
            } else {
                t[e] = static_cast<float>(acc[0][r + e] * kB + acc[1][r + e]);
                if (NDIG == 2) t[e] = t[e] * sc;
                if (NDIG == 3) t[e] = __builtin_fmaf(t[e], kB * sc, static_cast<float>(acc[2][r + e]) * sc);
                if (NDIG == 4) t[e] = __builtin_fmaf(t[e], kB * kB * sc, static_cast<float>(acc[2][r + e] * kB + acc[3][r + e]) * sc);
            }
        }
        x0 = relu_clamp(t[0]);
        x1 = relu_clamp(t[1]);
    };
    auto conv = [&](v16i (&acc)[NDIG], const v4i (&bfr)[6]) {
#pragma unroll
        for (int d = 0; d + 1 < NDIG; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][r] = 0;
        acc[NDIG - 1] = CONV_LDS ? *cbeta_lds : cbeta;
        if (CONV_LDS) {
            // weight digits stream through a three-deep register ring, two kernel-column steps ahead of their
            // MFMAs; the group barriers pin that order (left alone, hipcc issues each read right before its
            // use and waits out the LDS latency 6 x digits times per position)
            constexpr int RD = NDIG == 4 ? 2 : 3;                     // ring depth the 256-register budget allows
            v4i wq[RD][NDIG];
#pragma unroll
            for (int g = 0; g < RD - 1; ++g)
#pragma unroll
                for (int d = 0; d < NDIG; ++d) wq[g][d] = wcl[(g * NDIG + d) * 64];
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                if (ks + RD - 1 < 6) {
#pragma unroll
                    for (int d = 0; d < NDIG; ++d) wq[(ks + RD - 1) % RD][d] = wcl[((ks + RD - 1) * NDIG + d) * 64];
                }
#pragma unroll
                for (int d = 0; d < NDIG; ++d)
                    acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wq[ks % RD][d], bfr[ks], acc[d], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, (RD - 1) * NDIG, 0);      // DS reads of the leading steps
#pragma unroll
            for (int ks = 0; ks + RD - 1 < 6; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x100, NDIG, 0);             // reads of step ks + RD - 1
                __builtin_amdgcn_sched_group_barrier(0x008, NDIG, 0);             // MFMAs of step ks
            }
            __builtin_amdgcn_sched_group_barrier(0x008, (RD - 1) * NDIG, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int d = 0; d < NDIG; ++d)
                    acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wA[CONV_LDS ? 0 : ks][d], bfr[ks], acc[d], 0, 0, 0);
        }
    };

    // FOLD: groups whose counters this wavefront bumped after its last actor block (lane j < fold_n: group, the counter's value
    // before the add -- requested there, looked at only when the staging area is next free -- and the count that completes it)
    int fold_g = 0, fold_old = 0, fold_exp = 0, fold_n = 0;
    long long fold_r0 = 0, fold_r1 = -1;                              // rows of the last actor block whose arrival is still to be announced
    auto fold_drain = [&]() {                                         // call only where this wavefront's staging area is free
#ifdef OC_DIAGNOSTICS
        if (!FOLD || fold_n == 0) return;
        unsigned long long mine = __ballot(lane < fold_n && fold_old + 1 == fold_exp);
        fold_n = 0;
        if (!mine) return;
        // (no fence: the env body reads the actions with agent-scope loads, which do not trust this XCD's L2 -- an acquire fence here
        // would invalidate it whole, a release fence on the writers' side write it back whole: measured 381 us per step against 62)
        while (mine) {
            const int j = __ffsll(static_cast<long long>(mine)) - 1;
            mine &= mine - 1;
            if (a.fold_nt & 2) { if (lane == 0) a.fold_count[__builtin_amdgcn_readlane(fold_g, j)] = 0; continue; }   // timing diagnostics (OC_FOLD_DEBUG=2): no env step
            fold_env_group(a.fold_args, FoldStep{a.fold_actions, a.fold_obs, a.fold_reward, a.fold_done}, __builtin_amdgcn_readlane(fold_g, j), a.fold_nt,
                           static_cast<int>((ldsw - lds) >> 2), lane, a.fold_count);
            if (a.fold_nt & 4)                                        // timing diagnostics (OC_FOLD_DEBUG=4): the same group again, code now warm
                fold_env_group(a.fold_args, FoldStep{a.fold_actions, a.fold_obs, a.fold_reward, a.fold_done}, __builtin_amdgcn_readlane(fold_g, j), a.fold_nt,
                               static_cast<int>((ldsw - lds) >> 2), lane, a.fold_count);
        }
        wave_lds_sync();
#endif
    };
    auto fold_arrive = [&](long long r0, long long r1) {              // after a block's actions are stored
#ifdef OC_DIAGNOSTICS
        if (!FOLD) return;
        int g[4];
        const int n = fold_block_groups(r0, r1, a.fold_kitchens, g);
        const int mine = lane == 0 ? g[0] : lane == 1 ? g[1] : lane == 2 ? g[2] : g[3];
        if (lane < n) {
            fold_g = mine;
            fold_exp = a.fold_expected[mine];
            fold_old = atomicAdd(a.fold_count + mine, 1);
        }
        fold_n = n;
#else
        (void)r0; (void)r1;
#endif
    };
    (void)fold_g; (void)fold_old; (void)fold_exp; (void)fold_n;

    // cooperative blocks by flags (eight-wavefront builds): the hand-shake words start at zero -- written long before their first use,
    // which lies behind the network barrier every wavefront passes -- and count this workgroup's stretches that share a block
    int coop_seq = 0;
    if (CONV_LDS && a.lds_flag_offset && wave == 0 && lane < 16) reinterpret_cast<int *>(lds + a.lds_flag_offset)[lane] = 0;
  for (long long cur = wg0; cur < wg1;) {
    // (read the way policy_find_stretch read it -- scalar for tables mapped to workgroups, vector for a table split evenly -- so that it hits the cache
    // that read filled: the scalar and the vector L1 do not share lines)
    const int4 tile = a.tile_per_wg ? load_tile(a.tiles, ti) : a.tiles[ti];
    if (MULTI) {
        cur_slot = __builtin_amdgcn_readfirstlane((tile.w >> 16) & (kMaxSlots - 1));
        const PolicySlot &T = a.slot[cur_slot];
        RB = T.RB; P = T.L.P; PH = T.L.PH; H = T.L.H;
        colstep = H * OC_POLICY_C;
        rowp = ldsw + col * RB + 16 * h;
        wstart = static_cast<int>(T.L.off_b1);
    }
    const PolicySlot &S = a.slot[MULTI ? cur_slot : 0];
    const int tnb = policy_tile_blocks<GATHER>(S, tile);
    if (tnb == 0) { ++ti; continue; }
    const long long run_end = wg1 < tstart + tnb ? wg1 : tstart + tnb;
    const int first = static_cast<int>(cur - tstart), last = static_cast<int>(run_end - tstart);   // this stretch of the tile
    // a gather tile {list, class, net, flags} evaluates the rows listed for that class of that list (oc_partition_rows: class 0's ids
    // first, class 1's behind them); its rows are staged padded to whole 16-byte pieces
    const bool gather = GATHER && (tile.w & OC_POLICY_TILE_GATHER) != 0;
    const int32_t *gidx = gather ? S.row_index + static_cast<long long>(tile.x) * S.gather_len + ((tile.y & 1) ? policy_gather_rows(S, make_int4(tile.x, 0, 0, 0)) : 0)
                                 : nullptr;
    const long long row0 = gather ? 0 : tile.x;
    const int nrows = gather ? policy_gather_rows(S, tile) : static_cast<int>(row0 + tile.y > S.n_rows ? S.n_rows - row0 : tile.y);
    const int net = min(max(tile.z, 0), S.n_nets - 1);
    const int flags = gather ? (tile.w & ~0xff00) : tile.w;           // (a gather tile needs no select code: it only has its own rows)
    const int RS = gather ? (RB + 15) & ~15 : RB;                     // LDS row stride
    rowp = ldsw + col * RS + 16 * h;
    cur = run_end;
    if (run_end == tstart + tnb) { tstart += tnb; ++ti; }
    int myrow_next = 0;                                               // gather: row id of lane & 31 in the block being staged
    auto stage = [&](int b) -> int {                                  // (returns the vector-memory instructions issued, -1: unknown)
        int n = -1;
        if (gather) myrow_next = stage_rows_gather(S, ldsw, gidx, b * kRowsPerBlock, nrows, lane, n);
        else n = stage_rows(S, ldsw, row0 + static_cast<long long>(b) * kRowsPerBlock, lane);
        return n;
    };

    // this wavefront's first block of the stretch: in flight while (if) the weights are fetched
    if (loaded_net < 0) stamp(a, wave, lane, 10);
    // The first network of a launch.  Every wavefront of the CU wants its first rows AND its share of the weights at once: 150 KB
    // (cramped_room, eight wavefronts) through one CU's vector-memory path, served in order, rows from HBM -- requested all together,
    // nobody's rows are in before ~4 us and the matrix pipes idle until then.  So the second wavefront of each SIMD requests its rows
    // BEHIND its share of the weights (and waits for the weights only, with a counted vmcnt): the first four start computing when
    // 98 KB have arrived, and their SIMD partners' rows land under that work.  (One wavefront per SIMD, or a network change later in
    // the launch: rows first -- they come from furthest away, and a wavefront that arrives early has them in flight while it waits.)
    const bool reload = net != loaded_net || (MULTI && cur_slot != loaded_slot);
    const bool late_rows = kLateRows && reload && loaded_net < 0 && nwaves > 4 && wave >= 4;
    // ---- cooperative blocks.  A workgroup's wavefronts take the stretch's blocks round-robin, so a stretch of 4 k + 1 blocks on four
    // wavefronts costs k + 1 rounds although the last round is one block on one SIMD (config 4's per-GPU share: 102 of 256 workgroups
    // have five blocks, and the launch is two blocks long for everybody).  The remainder -- one block per four wavefronts, or two on a
    // four-wavefront workgroup -- is therefore SHARED: the wavefronts of a group (one per SIMD) split the block's conv positions
    // (a run of consecutive positions each), accumulate partial fc1 sums over their positions from the group leader's staged rows, and hand
    // them to the leader through their own (idle) staging areas; the leader adds them up and runs the dense tail alone.  Three
    // workgroup barriers per stretch, only here; every wavefront of the workgroup passes them.
    const int nb_stretch = last - first, rem = nb_stretch & (nwaves - 1);   // (a workgroup is 1, 4 or 8 wavefronts: no runtime division in the prologue)
    // (only in the builds that run one wavefront per SIMD -- conv digits in registers -- and not in the gather build, which is at its
    // register limit without a second block body.  Two wavefronts per SIMD, measured: the barrier in front of the shared block waits
    // for the slowest of eight drifting wavefronts, which costs more than the block -- 52.4 against 48.6 us for the bench line's launch)
    // Eight wavefronts (conv digits in LDS): the same split among wavefronts 0-3 only -- one per SIMD, the ones a stretch's 8 k + 1st / 2nd
    // block falls to; their SIMD partners 4-7 are a few us behind and take no part -- with LDS hand-shake words instead of the three
    // barriers (kCoopFlags; round 6: a barrier there waits for the slowest of eight drifting wavefronts, which cost more than the block).
    constexpr bool kCoopBuild = !FOLD && !GATHER && TAIL_LDS;
    constexpr bool kCoopFlags = CONV_LDS;
    const int gsz = (!kCoopBuild || !a.coop || rem == 0 || ((flags >> 8) & 0xff) || a.lds_wave_bytes < 8192) ? 0
                    // (measured, profiles/r06_policy_coop_variants.log: two leftover blocks as two groups of four -- wavefronts 4-7 sharing the second --
                    // cost the bench line's launch 2.7 us: the partners arrive late; a stretch of two or three blocks in all is faster unshared,
                    // 11.9 against 12.6 us for 16 368 evaluations; a stretch of ONE block is shared: 1000 evaluations 12.0 -> 11.2 us)
                    : kCoopFlags ? ((nwaves != 8 || a.lds_flag_offset == 0 || (nb_stretch < nwaves && nb_stretch != a.coop1)) ? 0 : rem == 1 ? 4 : rem == 2 ? a.coop2 : 0)
                    : nwaves != 4 ? 0 : rem == 1 ? 4 : rem == 2 ? 2 : 0;
    const bool coop = gsz > 0 && P >= gsz;
    coop_seq += coop ? 1 : 0;
    // the leader also runs fc2, head and the epilogue -- about three positions' worth of time -- while the others are already idle:
    // its run of positions is shorter by that much (L + 3 = (P - L) / (g - 1)), one position at least
    const int gshift = gsz == 4 ? 2 : 1;                              // (gsz is 2 or 4: x / gsz as a shift; a negative numerator ends at the max below either way)
    const int coop_lead = coop ? min(max((P - kCoopTailPositions * (gsz - 1)) >> gshift, 1), P - (gsz - 1)) : 0;
    const int last_full = coop ? last - rem : last;                   // blocks [first, last_full): one wavefront each
    const int grp = coop ? wave >> gshift : 0, member = coop ? wave - grp * gsz : 0;
    const bool in_coop = coop && grp < rem, leader = in_coop && member == 0;
    const int first_item = first + wave < last_full ? first + wave : leader ? last_full + grp : -1;
    const bool have_rows = first_item >= 0;
    int rows_pending = 0;                                             // row instructions issued behind the weight copy: they need not have landed at the barrier
    const uint8_t *blob = nullptr;
    int tail_bytes = 0;
    bool weights_in_lds = false;
    int hdr_n_out = 0;
    int hdr_d1 = 0, hdr_d2 = 0, hdr_d3 = 0;
    // (two passes through ONE copy of the row-staging code -- early rows, weights, late rows; a second inlined copy of stage_rows
    // costs the 256-register build SGPR spills it has no lane left for)
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (have_rows && late_rows == (pass == 1)) {
            const int n = stage(first_item);
            if (pass == 1) rows_pending = n;
        }
        if (pass == 1 || !reload) continue;
        if (loaded_net < 0) stamp(a, wave, lane, 11);
        if (loaded_net >= 0) __syncthreads();                         // everybody is done with the previous network
        blob = S.packed + static_cast<size_t>(net) * S.L.total;
        if (a.lds_weight_bytes == 0) {                                // (wave tiles: requested here, used behind the barrier)
            hdr_n_out = reinterpret_cast<const int *>(blob)[4];
            const int *dg = reinterpret_cast<const int *>(blob + S.L.off_ds);
            hdr_d1 = dg[0]; hdr_d2 = dg[1]; hdr_d3 = dg[2];
        }
        // LDS-resident weights, verbatim pieces of the image: [conv digits + scale/bias] when CONV_LDS, then the
        // tail from fc1 (or, when fc1 does not fit, from b1) to the end, each padded to whole KB in LDS.  LDS-DMA: every piece is
        // in flight at once (a load / wait / ds_write loop cost one L2 round trip per pass); see copy_to_lds.  The lanes past the
        // tail's end bring the image HEADER along: the network's own output count (header word 4) is then an LDS read behind the
        // barrier instead of a scalar load that was waited for on the spot, one more round trip in front of the copy.
        tail_bytes = static_cast<int>(S.L.off_ds + 32 - wstart);
        weights_in_lds = a.lds_weight_bytes > 0;
        if (weights_in_lds) {
            // (the wavefront index behind an empty asm: everything copy_to_lds derives from it is then computed here, where it is used --
            // hoisted out of the tile loop those values stayed live through the block loop, as SGPRs spilled to lanes of one more VGPR
            // than the 256-register build has to spare)
            int wv = wave;
            asm volatile("" : "+s"(wv));
            if (CONV_LDS) copy_to_lds(blob + S.L.off_conv, lds, (conv_bytes + 1023) >> 10, 64, blob, wv, nwaves, lane);   // (its last KB runs on into the image: read, never used)
            const int nk = (tail_bytes + 32 + 1023) >> 10;
            copy_to_lds(blob + wstart, lds_tail, nk, (tail_bytes >> 4) - (nk - 1) * 64, blob, wv, nwaves, lane);
        }
        if (!CONV_LDS) {
            const v4i *src = reinterpret_cast<const v4i *>(blob + S.L.off_conv);
#pragma unroll
            for (int ks = 0; ks < (CONV_LDS ? 1 : 6); ++ks)
#pragma unroll
                for (int d = 0; d < NDIG; ++d) wA[ks][d] = src[(ks * NDIG + d) * 64 + lane];
            cbeta = *reinterpret_cast<const v16i *>(blob + S.L.off_conv_sb + 128 + h * 64);
        }
        w1g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(blob + S.L.off_fc1), 0, P * OC_POLICY_FC1_POS_BYTES, 0x00020000);
        if (!TAIL_LDS) {                                              // the tail stays in global memory (L2): this network's sections
            b1 = reinterpret_cast<const float *>(blob + S.L.off_b1) + h * 16;
            w2 = reinterpret_cast<const v4i *>(blob + S.L.off_fc2) + lane;
            b2 = reinterpret_cast<const float *>(blob + S.L.off_b2) + h * 16;
            wh = reinterpret_cast<const v4i *>(blob + S.L.off_head) + (8 * h + (col & 7));
            bh = reinterpret_cast<const float *>(blob + S.L.off_bh) + h * 4;
        }
    }
    if (reload) {
        if (loaded_net < 0) stamp(a, wave, lane, 12);
        wait_vmcnt(rows_pending);                                     // this wavefront's share of the weight copy has landed
        if (loaded_net < 0) stamp(a, wave, lane, 13);
        // the network is in LDS.  A bare barrier (__syncthreads() would first wait for vmcnt(0), i.e. for the rows just requested): an
        // LDS-DMA write is complete when the issuing wavefront's vmcnt has counted it, and every wavefront has waited for its share
        asm volatile("s_barrier" ::: "memory");
        n_out = weights_in_lds ? __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(lds_tail + tail_bytes + 16))   // header word 4, copied behind the tail
                               : hdr_n_out;                                                                                   // (wave tiles: nothing of the network is in LDS)
        n_out = min(n_out, kMaxOut);
        {
            // the image's descale section (three words, read one by one: as one 16-byte vector behind a select hipcc 7.2 used word 0 for all three)
            const int *dsp = reinterpret_cast<const int *>(lds_tail + tail_bytes - 32);
            const int d1 = weights_in_lds ? dsp[0] : hdr_d1;
            const int d2 = weights_in_lds ? dsp[1] : hdr_d2;
            const int d3 = weights_in_lds ? dsp[2] : hdr_d3;
            ds1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(d1));
            ds2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(d2));
            ds3 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(d3));
        }
        loaded_net = net;
        loaded_slot = cur_slot;
        stamp(a, wave, lane, 1);
        if (a.desync > 0 && wave >= 4)                                // experiment: break the lock-step of a SIMD's two waves
            for (int i = 0; i < a.desync; i += 100) __builtin_amdgcn_s_sleep(100);
    }

    for (int blk = first + wave; blk < last_full; blk += nwaves) {
#define OC_BLOCK_COOP 0
#include "policy_block_body.h"
#undef OC_BLOCK_COOP
    }
    if constexpr (kCoopBuild) {
        if (coop) {
            const int blk = last_full + min(grp, rem - 1);           // (wavefronts outside the groups only pass the barriers)
            do {
#define OC_BLOCK_COOP 1
#include "policy_block_body.h"
#undef OC_BLOCK_COOP
            } while (false);
        }
    }
  }
#ifdef OC_DIAGNOSTICS
    if (FOLD) {
        fold_drain();                                                 // requests of the last block but one
        if (fold_r1 >= 0) {                                           // the last block: its actions (agent-scope stores, written through) must have completed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fold_arrive(fold_r0, fold_r1);
            fold_drain();
        }
        // groups no actor block touches (a table that leaves kitchens to actions written elsewhere): stepped here, spread over all wavefronts
        const int total_waves = gridDim.x * nwaves, me = blockIdx.x * nwaves + wave;
        for (int g = me; g < a.fold_groups; g += total_waves)
            if (a.fold_expected[g] == 0) {
                fold_env_group(a.fold_args, FoldStep{a.fold_actions, a.fold_obs, a.fold_reward, a.fold_done}, g, a.fold_nt, static_cast<int>((ldsw - lds) >> 2), lane, a.fold_count);
                wave_lds_sync();
            }
    }
#endif
    stamp(a, wave, lane, 8); stamp(a, wave, lane, 15);
}

#include "policy_duo_kernel.h"

// The three dense layers' power-of-two exponents (their largest |weight| into [2^13, 2^14), policy_pack.h) -> the image's descale
// section: one workgroup, first in the stream; the pack kernel below reads the exponents from there.
__global__ void oc_policy_scales_kernel(oc_policy_layout L, oc_policy_weights w, uint32_t *dst) {
    __shared__ float cs[OC_POLICY_O];
    __shared__ float red[3][256];
    const int t = threadIdx.x;
    if (t < OC_POLICY_O) cs[t] = oc_policy_conv_scale(w, t, L.digits) * oc_pow2f(oc_policy_conv_shift(L.digits));
    __syncthreads();
    float m1 = 0.f, m2 = 0.f, m3 = 0.f;
    auto upd = [](float &m, float v) { v = v < 0.f ? -v : v; if (v > m && v < 3.0e38f) m = v; };
    const int per_out = OC_POLICY_O * L.P;
    for (int i = t; i < OC_POLICY_HID * per_out; i += 256) upd(m1, w.fc1_w[i] * cs[(i % per_out) / L.P]);
    for (int i = t; i < OC_POLICY_HID * OC_POLICY_HID; i += 256) upd(m2, w.fc2_w[i]);
    for (int i = t; i < L.n_out * OC_POLICY_HID; i += 256) upd(m3, w.head_w[i]);
    red[0][t] = m1; red[1][t] = m2; red[2][t] = m3;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (t < d)
            for (int k = 0; k < 3; ++k) red[k][t] = red[k][t] > red[k][t + d] ? red[k][t] : red[k][t + d];
        __syncthreads();
    }
    if (t < 8) dst[L.off_ds / 4 + t] = oc_policy_pack_ds(oc_policy_scales_from_max(red[0][0], red[1][0], red[2][0]), t);
}

__global__ void oc_policy_pack_kernel(oc_policy_layout L, oc_policy_weights w, uint32_t *dst) {
    const long long words = L.total / 4, ds0 = L.off_ds / 4;
    oc_policy_scales sc;
    sc.e1 = static_cast<int32_t>(dst[ds0 + 4]); sc.e2 = static_cast<int32_t>(dst[ds0 + 5]); sc.e3 = static_cast<int32_t>(dst[ds0 + 6]);
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < words;
         i += static_cast<long long>(gridDim.x) * blockDim.x)
        if (i < ds0 || i >= ds0 + 8) dst[i] = oc_policy_pack_word(L, w, sc, i);     // (the descale section is the scales kernel's)
}

unsigned long long *g_trace = nullptr;    // device buffer of the diagnostics entry below (stays NULL in the product library)
#ifdef OC_DIAGNOSTICS
long long g_fold_launches = 0;            // policy launches that carried the env step (oc_debug_fold_launches)
#endif

int check_shape(const oc_policy_shape *s, const char *who) {
    if (!s) return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: shape is NULL", who);
    if (s->width < 3 || s->height < 3 || s->width * s->height > OC_MAX_TILES)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: grid %dx%d outside 3x3 .. %d tiles", who, s->width, s->height, OC_MAX_TILES);
    if (s->hidden != OC_POLICY_HID)
        return oc_internal_fail(OC_ERR_UNSUPPORTED, "%s: hidden_size %d (the fused kernel is built for %d, the reference's training setting)",
                                who, s->hidden, OC_POLICY_HID);
    if (s->n_out < 1 || s->n_out > kMaxOut) return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: n_out %d outside 1..%d", who, s->n_out, kMaxOut);
    if (s->digits < 1 || s->digits > 4) return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: digits %d outside 1..4", who, s->digits);
    return OC_OK;
}

bool any_null(const oc_policy_weights &w) {
    return !w.conv_w || !w.conv_b || !w.fc1_w || !w.fc1_b || !w.fc2_w || !w.fc2_b || !w.head_w || !w.head_b;
}

}  // namespace

extern "C" {

int64_t oc_policy_packed_bytes(const oc_policy_shape *shape) {
    if (check_shape(shape, "oc_policy_packed_bytes") != OC_OK) return -1;
    return oc_policy_make_layout(shape->width, shape->height, shape->digits, shape->n_out).total;
}

int oc_policy_pack_host(const oc_policy_shape *shape, const float *conv_w, const float *conv_b, const float *fc1_w,
                        const float *fc1_b, const float *fc2_w, const float *fc2_b, const float *head_w, const float *head_b,
                        void *host_dst) {
    if (int rc = check_shape(shape, "oc_policy_pack_host")) return rc;
    const oc_policy_weights w = {conv_w, conv_b, fc1_w, fc1_b, fc2_w, fc2_b, head_w, head_b};
    if (any_null(w) || !host_dst) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_pack_host: NULL pointer");
    const oc_policy_layout L = oc_policy_make_layout(shape->width, shape->height, shape->digits, shape->n_out);
    uint32_t *dst = static_cast<uint32_t *>(host_dst);
    const oc_policy_scales sc = oc_policy_compute_scales(L, w);
    for (int64_t i = 0; i < L.total / 4; ++i) dst[i] = oc_policy_pack_word(L, w, sc, i);
    return OC_OK;
}

int oc_policy_pack(const oc_policy_shape *shape, const float *d_conv_w, const float *d_conv_b, const float *d_fc1_w,
                   const float *d_fc1_b, const float *d_fc2_w, const float *d_fc2_b, const float *d_head_w,
                   const float *d_head_b, void *d_dst, void *stream) {
    if (int rc = check_shape(shape, "oc_policy_pack")) return rc;
    const oc_policy_weights w = {d_conv_w, d_conv_b, d_fc1_w, d_fc1_b, d_fc2_w, d_fc2_b, d_head_w, d_head_b};
    if (any_null(w) || !d_dst) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_pack: NULL pointer");
    if (reinterpret_cast<uintptr_t>(d_dst) & 15) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_pack: d_dst must be 16-byte aligned");
    const oc_policy_layout L = oc_policy_make_layout(shape->width, shape->height, shape->digits, shape->n_out);
    const unsigned grid = static_cast<unsigned>((L.total / 4 + 255) / 256);
    const OcDeviceScope on_device(oc_device_of_pointer(d_dst));
    hipLaunchKernelGGL(oc_policy_scales_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), L, w, static_cast<uint32_t *>(d_dst));
    hipLaunchKernelGGL(oc_policy_pack_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), L, w,
                       static_cast<uint32_t *>(d_dst));
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return oc_internal_fail(OC_ERR_HIP, "oc_policy_pack: launch failed: %s", hipGetErrorString(err));
    return OC_OK;
}

}  // extern "C"

namespace {

// seg_slot / slot_cost: NULL for a single grid; otherwise the slot of each segment and the relative cost of one 32-row block of
// each slot (the balanced plan then evens out cost, not block counts, and the tiles carry the slot in their flags).
int64_t plan_tiles_impl(const char *who, const int64_t *seg_row0, const int64_t *seg_rows, const int32_t *seg_net, const int32_t *seg_flags,
                        const int32_t *seg_slot, const float *slot_cost, int32_t n_slots, int32_t n_segs, int32_t rows_per_tile,
                        int32_t *host_tiles, int64_t capacity) {
    const bool balanced = rows_per_tile < 0;                          // -rows_per_tile = number of workgroups to fill
    if (!seg_row0 || !seg_rows || !seg_net || n_segs < 0 || rows_per_tile == 0 ||
        (!balanced && (rows_per_tile < kRowsPerBlock || rows_per_tile % kRowsPerBlock)))
        return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: bad arguments (rows_per_tile: a multiple of %d, or -workgroups)", who, kRowsPerBlock);
    std::vector<int64_t> blocks(n_segs), pieces(n_segs);
    std::vector<double> cost(n_segs, 1.0);
    double total = 0;
    for (int32_t g = 0; g < n_segs; ++g) {
        if (seg_row0[g] < 0 || seg_rows[g] < 0 || seg_row0[g] + seg_rows[g] > INT32_MAX)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: segment %d out of range", who, g);
        if (seg_slot) {
            if (seg_slot[g] < 0 || seg_slot[g] >= n_slots || !(slot_cost[seg_slot[g]] > 0.f))
                return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: segment %d names slot %d (of %d) or a non-positive cost", who, g, seg_slot[g], n_slots);
            cost[g] = slot_cost[seg_slot[g]];
        }
        blocks[g] = (seg_rows[g] + kRowsPerBlock - 1) / kRowsPerBlock;
        total += blocks[g] * cost[g];
    }
    if (balanced) {
        // One tile = one workgroup = one network, so no workgroup reloads weights.  Pieces per segment in proportion
        // to its work, then the spare workgroups go, one at a time, to the segment with the most expensive pieces.
        const int64_t wgs = -static_cast<int64_t>(rows_per_tile);
        int64_t used = 0;
        for (int32_t g = 0; g < n_segs; ++g) {
            pieces[g] = blocks[g] == 0 ? 0 : std::max<int64_t>(1, std::min(blocks[g], total > 0 ? static_cast<int64_t>(wgs * (blocks[g] * cost[g]) / total) : 1));
            used += pieces[g];
        }
        // A workgroup's wavefronts take its blocks round-robin, so what a piece costs is its number of ROUNDS: 17 blocks on 8
        // wavefronts take as long as 24.  Try piece sizes that are whole rounds (multiples of 8 blocks: whole rounds for 4
        // wavefronts too): the smallest such size whose pieces fit the workgroups, if it needs fewer rounds than the
        // proportional split (15 segments of 4369 kitchens per side on 256 workgroups: 137-block segments got 8 pieces of
        // 17-18 blocks, three rounds; nine pieces of 15-16 are two).
        if (!seg_slot) {
            int64_t worst_now = 0, all_blocks = 0;
            for (int32_t g = 0; g < n_segs; ++g) {
                if (pieces[g] > 0) worst_now = std::max(worst_now, (blocks[g] + pieces[g] - 1) / pieces[g]);
                all_blocks += blocks[g];
            }
            for (int64_t size = std::max<int64_t>(8, (all_blocks / wgs + 7) / 8 * 8); size < worst_now; size += 8) {
                int64_t need = 0;
                for (int32_t g = 0; g < n_segs; ++g) need += (blocks[g] + size - 1) / size;
                if (need > wgs) continue;
                if ((size + 7) / 8 < (worst_now + 7) / 8) {
                    used = 0;
                    for (int32_t g = 0; g < n_segs; ++g) { pieces[g] = (blocks[g] + size - 1) / size; used += pieces[g]; }
                }
                break;
            }
        }
        for (; used < wgs; ++used) {
            int32_t best = -1;
            double worst = 0.0;                                       // only segments whose pieces exceed one block
            for (int32_t g = 0; g < n_segs; ++g)
                if (pieces[g] > 0 && pieces[g] < blocks[g] && static_cast<double>(blocks[g]) / pieces[g] * cost[g] > worst) {
                    worst = static_cast<double>(blocks[g]) / pieces[g] * cost[g];
                    best = g;
                }
            if (best < 0) break;
            ++pieces[best];
        }
    } else {
        for (int32_t g = 0; g < n_segs; ++g) pieces[g] = (seg_rows[g] + rows_per_tile - 1) / rows_per_tile;
    }
    int64_t n = 0;
    for (int32_t g = 0; g < n_segs; ++g) {
        int64_t done_blocks = 0;
        for (int64_t i = 0; i < pieces[g]; ++i, ++n) {
            const int64_t nb = balanced ? blocks[g] / pieces[g] + (i < blocks[g] % pieces[g] ? 1 : 0) : rows_per_tile / kRowsPerBlock;
            const int64_t r = done_blocks * kRowsPerBlock, left = seg_rows[g] - r;
            done_blocks += nb;
            if (!host_tiles) continue;
            if (n >= capacity) return oc_internal_fail(OC_ERR_INVALID_ARG, "%s: capacity %lld too small", who, (long long)capacity);
            host_tiles[4 * n + 0] = static_cast<int32_t>(seg_row0[g] + r);
            host_tiles[4 * n + 1] = static_cast<int32_t>(std::min(left, nb * kRowsPerBlock));
            host_tiles[4 * n + 2] = seg_net[g];
            host_tiles[4 * n + 3] = ((seg_flags ? seg_flags[g] : 0) & 0xFFFF) | (seg_slot ? seg_slot[g] << 16 : 0);
        }
    }
    return n;
}

}  // namespace

extern "C" {

int64_t oc_policy_plan_tiles(const int64_t *seg_row0, const int64_t *seg_rows, const int32_t *seg_net, const int32_t *seg_flags,
                             int32_t n_segs, int32_t rows_per_tile, int32_t *host_tiles, int64_t capacity) {
    return plan_tiles_impl("oc_policy_plan_tiles", seg_row0, seg_rows, seg_net, seg_flags, nullptr, nullptr, 1, n_segs, rows_per_tile, host_tiles, capacity);
}

int64_t oc_policy_plan_tiles_multi(const int64_t *seg_row0, const int64_t *seg_rows, const int32_t *seg_net, const int32_t *seg_flags,
                                   const int32_t *seg_slot, const float *slot_cost, int32_t n_slots, int32_t n_segs, int32_t workgroups,
                                   int32_t *host_tiles, int64_t capacity, int32_t *tiles_per_workgroup, int32_t *waves_per_workgroup) {
    if (!seg_row0 || !seg_rows || !seg_net || !seg_slot || !slot_cost || !tiles_per_workgroup || !waves_per_workgroup || n_slots < 1 ||
        n_slots > kMaxSlots || workgroups < 1 || n_segs < 0)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_plan_tiles_multi: NULL pointer, 1..%d slots with their costs, workgroups >= 1", kMaxSlots);
    // At least two workgroups per segment: the single-grid plan with costs -- workgroups shared out over the segments in proportion
    // to their work, one tile each, no workgroup ever changes network (config 3's actors: 96 us per joint step against 103 packed).
    // With fewer, whole workgroups per segment no longer balance (235 segments on 256 workgroups: 196 us against 155 packed).
    const int32_t waves_hint = *waves_per_workgroup;                  // in: 0 = choose, 1 = wave tiles, 4 = four-wavefront workgroups
    if (waves_hint != 0 && waves_hint != 1 && waves_hint != 4)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_plan_tiles_multi: *waves_per_workgroup is 0 (choose), 1 or 4 on entry");
    *waves_per_workgroup = 4;
    {
        int32_t live = 0;
        int64_t all_blocks = 0;
        for (int32_t g = 0; g < n_segs; ++g) {
            live += seg_rows[g] > 0 ? 1 : 0;
            all_blocks += seg_rows[g] > 0 ? (seg_rows[g] + kRowsPerBlock - 1) / kRowsPerBlock : 0;
        }
        // Short segments (fewer than kWaveTileBlocks 32-row blocks each, on average): wave tiles -- four one-wavefront workgroups
        // per CU, each with its own network in L2 / registers, planned as 4 x workgroups packed stretches.
        const bool wave_tiles = waves_hint ? waves_hint == 1 : (live > 0 && all_blocks < static_cast<int64_t>(kWaveTileBlocks) * live);
        if (wave_tiles) {
            *waves_per_workgroup = 1;
            workgroups *= 4;
        } else if (2 * live <= workgroups) {
            *tiles_per_workgroup = 1;
            return plan_tiles_impl("oc_policy_plan_tiles_multi", seg_row0, seg_rows, seg_net, seg_flags, seg_slot, slot_cost, n_slots, n_segs,
                                   -workgroups, host_tiles, capacity);
        }
    }
    // Otherwise: the 32-row blocks of all segments, in order, form one sequence with a cost each; workgroup w takes the blocks whose
    // cost midpoint falls into [w, w + 1) x (total cost / workgroups): equal work whatever the number and size of the segments
    // (one tile per workgroup, as the single-grid plan has it, stops balancing once there are more segments than workgroups:
    // config 5 has 1260).  A segment that spans a boundary is cut there; a workgroup whose stretch runs through several segments
    // reloads the weights at each.  The table holds `*tiles_per_workgroup` entries per workgroup, padded with empty tiles.
    std::vector<int64_t> blocks(n_segs);
    std::vector<double> cost(n_segs);
    double total = 0;
    for (int32_t g = 0; g < n_segs; ++g) {
        if (seg_row0[g] < 0 || seg_rows[g] < 0 || seg_row0[g] + seg_rows[g] > INT32_MAX)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_plan_tiles_multi: segment %d out of range", g);
        if (seg_slot[g] < 0 || seg_slot[g] >= n_slots || !(slot_cost[seg_slot[g]] > 0.f))
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_plan_tiles_multi: segment %d names slot %d (of %d) or a non-positive cost", g, seg_slot[g], n_slots);
        cost[g] = slot_cost[seg_slot[g]];
        blocks[g] = (seg_rows[g] + kRowsPerBlock - 1) / kRowsPerBlock;
        total += blocks[g] * cost[g];
    }
    const double share = total > 0 ? total / workgroups : 1.0;
    auto owner = [&](double mid) { return std::min<int64_t>(workgroups - 1, static_cast<int64_t>(mid / share)); };
    struct Piece { int32_t seg; int64_t b0, nb; };
    std::vector<std::vector<Piece>> of_wg(workgroups);
    double x = 0;
    for (int32_t g = 0; g < n_segs; ++g) {
        int64_t j = 0;
        while (j < blocks[g]) {
            const int64_t w = owner(x + (j + 0.5) * cost[g]);
            int64_t e = j + 1;                                        // extend while the owner stays the same
            while (e < blocks[g] && owner(x + (e + 0.5) * cost[g]) == w) ++e;
            of_wg[w].push_back(Piece{g, j, e - j});
            j = e;
        }
        x += blocks[g] * cost[g];
    }
    size_t per = 1;
    for (const auto &v : of_wg) per = std::max(per, v.size());
    *tiles_per_workgroup = static_cast<int32_t>(per);
    const int64_t n = static_cast<int64_t>(per) * workgroups;
    if (!host_tiles) return n;
    if (n > capacity) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_plan_tiles_multi: capacity %lld too small", (long long)capacity);
    memset(host_tiles, 0, static_cast<size_t>(n) * 4 * sizeof(int32_t));
    for (int64_t w = 0; w < workgroups; ++w)
        for (size_t i = 0; i < of_wg[w].size(); ++i) {
            const Piece &pc = of_wg[w][i];
            int32_t *t = host_tiles + 4 * (w * static_cast<int64_t>(per) + static_cast<int64_t>(i));
            const int64_t r = pc.b0 * kRowsPerBlock;
            t[0] = static_cast<int32_t>(seg_row0[pc.seg] + r);
            t[1] = static_cast<int32_t>(std::min(seg_rows[pc.seg] - r, pc.nb * kRowsPerBlock));
            t[2] = seg_net[pc.seg];
            t[3] = ((seg_flags ? seg_flags[pc.seg] : 0) & 0xFFFF) | (seg_slot[pc.seg] << 16);
        }
    return n;
}

// (defined below; the CU-budget / per-engine-stream forms are reachable from outside only through the diagnostics library's oc_exp_* entry points)
static int policy_forward_cus(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                              const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                              float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                              const int32_t *d_row_select, int32_t cu_budget, void *stream);
static int group_policy_rollout_streams(oc_group *g, const oc_group_policy_slot *slots, void *const *policy_streams, const int32_t *cu_budgets,
                                        int64_t T, int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream);

int oc_policy_forward(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                      const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                      float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                      const int32_t *d_row_select, void *stream) {
    return policy_forward_cus(shape, d_packed, n_nets, d_obs, n_rows, d_tiles, n_tiles, d_raw, d_actions, d_action_logp, d_logp, sample_mode,
                              seed, counter, d_counter, d_row_select, 0, stream);
}

}  // extern "C"

namespace {

// Fill one slot of the launch arguments (argument checks shared by the single- and the multi-grid entry points).
int fill_slot(PolicySlot &t, const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
              float *d_raw, int32_t *d_actions, float *d_action_logp, float *d_logp, const int32_t *d_row_select, uint32_t seed) {
    if (int rc = check_shape(shape, "oc_policy_forward")) return rc;
    if (!d_packed || !d_obs || n_nets < 1 || n_rows < 1) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: NULL pointer or empty input");
    if ((reinterpret_cast<uintptr_t>(d_packed) & 15) || (reinterpret_cast<uintptr_t>(d_obs) & 3))
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: d_packed must be 16-byte aligned, d_obs 4-byte aligned");
    if (n_rows > INT32_MAX) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: more than 2^31 rows");
    t.packed = static_cast<const uint8_t *>(d_packed);
    t.L = oc_policy_make_layout(shape->width, shape->height, shape->digits, shape->n_out);
    t.n_nets = n_nets;
    t.RB = shape->width * shape->height * OC_POLICY_C;
    t.obs = d_obs;
    t.n_rows = n_rows;
    if (n_rows * t.RB < 16) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: input smaller than 16 bytes");
    t.raw = d_raw; t.actions = d_actions; t.action_logp = d_action_logp; t.logp = d_logp; t.row_select = d_row_select;
    t.row_index = nullptr; t.row_count = nullptr; t.gather_len = 0; t.gather_magic = 0;
    t.seed = seed;
    t.pad_ = 0;
    return OC_OK;
}

// Index lists for gather tiles (oc_partition_rows): n_lists lists of gather_len row ids with their two class counts.
int fill_gather(PolicySlot &t, const int32_t *d_row_index, const int32_t *d_row_count, int64_t gather_len) {
    if (!d_row_index || !d_row_count || gather_len < 1 || gather_len > t.n_rows)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_gather: NULL index / count arrays, or a list length outside 1 .. n_rows");
    t.row_index = d_row_index; t.row_count = d_row_count;
    t.gather_len = static_cast<int>(gather_len);
    const int ppr = (t.RB + 15) >> 4;                                  // 16-byte pieces per padded LDS row
    t.gather_magic = ((1 << 24) + ppr - 1) / ppr;                      // f / ppr == (f * magic) >> 24 for every piece index of a block
    return OC_OK;
}

int staging_bytes(int RB) { return (kRowsPerBlock * RB + 16 + 1023) / 1024 * 1024; }   // whole 1 KB LDS-DMA instructions (stage_rows)

// Pick the LDS plan and launch.  `multi`: the tile table runs over a.n_slots grids.
int launch_policy(PolicyArgs &a, int digits, bool multi, int32_t cu_budget, hipStream_t stream, bool wave_tiles = false) {
    const PolicySlot &s0 = a.slot[0];
    a.trace = g_trace;
    a.lds_wave_bytes = 0;
    for (int i = 0; i < a.n_slots; ++i) {
        a.lds_wave_bytes = std::max(a.lds_wave_bytes, staging_bytes(a.slot[i].RB));
        if (a.slot[i].row_index)                                       // gathered rows are staged padded to whole 16-byte pieces, in whole 1 KB instructions
            a.lds_wave_bytes = std::max(a.lds_wave_bytes, (kRowsPerBlock * ((a.slot[i].RB + 15) & ~15) + 1023) / 1024 * 1024);
    }
    constexpr long long kLds = 160 * 1024;
    const long long w_tail_all = s0.L.off_ds + 32 - s0.L.off_fc1, w_tail_small = s0.L.off_ds + 32 - s0.L.off_b1;
    const long long w_conv = s0.L.off_fc1 - s0.L.off_conv, obs1 = a.lds_wave_bytes;
    // Plan (profiles/r02_policy_plan_sweep.log):
    //   8 wavefronts (two per SIMD, 256 registers each) when eight staging areas fit next to the conv digits and the small
    //   tail: fc1 (12 KB per position) is then prefetched from L2 one position ahead (cramped_room: 102 us for 245 760
    //   evaluations against 105-107 with four wavefronts);
    //   otherwise 4 wavefronts (one per SIMD, 512 registers) with the conv digits in REGISTERS -- at one wavefront per SIMD the
    //   24 LDS fragment reads per position are latency nobody hides (counter_circuit 133 vs 147 us, forced_coordination 134
    //   vs 156) -- and fc1 LDS-resident if that fits, else from L2; then fewer wavefronts for very large rows.
    //   Several grids in one launch: the four-wavefront plan with fc1 in L2 (the LDS image is then the same for every grid).
    int waves = 8;
    bool fc1_lds = false, conv_lds = true;
    auto kb = [](long long b) { return (b + 1023) / 1024 * 1024; };   // either part is copied in whole 1 KB instructions, two header pieces behind the tail
    if (multi || kb(w_conv) + kb(w_tail_small + 32) + 8 * obs1 > kLds) {
        waves = 4;
        conv_lds = false;
        fc1_lds = !multi && kb(w_tail_all + 32) + 4 * obs1 <= kLds;
    }
    // Two cooperative plans for the grids where eight staging areas do not fit were built and measured in round 2 -- two
    // wavefronts of a SIMD sharing one staged block and splitting its conv positions, and a conv-wavefront / dense-wavefront
    // role split -- both parity-green and both 20-30 % SLOWER than one wavefront per SIMD (commits 'pair plan' / 'conv/dense
    // role split'; profiles/r02_policy_pair_split.log, r02_policy_conv_dense_split.log).  tools/probes/coexec.hip shows why:
    // on one gfx950 SIMD a partner wavefront's VALU work does not overlap back-to-back MFMAs of any type (both together cost
    // the sum, not the maximum), so a second wavefront per SIMD can only fill MFMA-free gaps, not hide work.
    const char *const env_plan = OC_DIAG_ENV("OC_POLICY_PLAN");           // diagnostics: "waves,fc1_lds,conv_lds" (read per call)
    if (env_plan && env_plan[0] && !multi) {
        int w = 0, f = 0, c = 0;
        if (sscanf(env_plan, "%d,%d,%d", &w, &f, &c) == 3 && w >= 1 && w <= 8) { waves = w; fc1_lds = f != 0; conv_lds = c != 0; }
    }
    a.lds_weight_bytes = static_cast<int>((conv_lds ? kb(w_conv) : 0) + kb((fc1_lds ? w_tail_all : w_tail_small) + 32));
    if ((fc1_lds ? w_tail_all : w_tail_small) % 1024 > 1024 - 32)     // (the two header pieces ride in the tail's last 1 KB instruction; 544 with this image layout)
        return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_policy_forward: image layout leaves no room for the header behind the LDS tail");
    if (wave_tiles) { waves = 1; a.lds_weight_bytes = 0; }           // one-wavefront workgroups, the whole network in L2 / registers
    while (waves > 1 && a.lds_weight_bytes + static_cast<long long>(waves) * a.lds_wave_bytes > kLds) waves >>= 1;
    size_t lds = static_cast<size_t>(a.lds_weight_bytes) + static_cast<size_t>(waves) * a.lds_wave_bytes;
    a.lds_flag_offset = 0;
    if (waves == 8 && conv_lds && !multi && !wave_tiles && lds + 64 <= static_cast<size_t>(kLds)) {   // hand-shake words of the cooperative blocks (oc_policy_kernel)
        a.lds_flag_offset = static_cast<int>(lds);
        lds += 64;
    }
    if (lds > static_cast<size_t>(kLds))
        return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_policy_forward: rows of %d bytes do not fit the LDS", a.lds_wave_bytes / kRowsPerBlock);
    typedef void (*kernel_t)(const int4 *, int, const PolicyArgs);
    static const kernel_t kernels[4][2][2] = {
        {{oc_policy_kernel<1, false, false>, oc_policy_kernel<1, false, true>}, {oc_policy_kernel<1, true, false>, oc_policy_kernel<1, true, true>}},
        {{oc_policy_kernel<2, false, false>, oc_policy_kernel<2, false, true>}, {oc_policy_kernel<2, true, false>, oc_policy_kernel<2, true, true>}},
        {{oc_policy_kernel<3, false, false>, oc_policy_kernel<3, false, true>}, {oc_policy_kernel<3, true, false>, oc_policy_kernel<3, true, true>}},
        {{oc_policy_kernel<4, false, false>, oc_policy_kernel<4, false, true>}, {oc_policy_kernel<4, true, false>, oc_policy_kernel<4, true, true>}}};
    static const kernel_t kernels_multi[4] = {oc_policy_kernel<1, false, false, true>, oc_policy_kernel<2, false, false, true>,
                                              oc_policy_kernel<3, false, false, true>, oc_policy_kernel<4, false, false, true>};
    static const kernel_t kernels_wave[4] = {oc_policy_kernel<1, false, false, true, false>, oc_policy_kernel<2, false, false, true, false>,
                                             oc_policy_kernel<3, false, false, true, false>, oc_policy_kernel<4, false, false, true, false>};
#define OC_PK(D, F, C) oc_policy_kernel<D, F, C, false, true, true>
    static const kernel_t kernels_gather[4][2][2] = {{{OC_PK(1, false, false), OC_PK(1, false, true)}, {OC_PK(1, true, false), OC_PK(1, true, true)}},
                                                     {{OC_PK(2, false, false), OC_PK(2, false, true)}, {OC_PK(2, true, false), OC_PK(2, true, true)}},
                                                     {{OC_PK(3, false, false), OC_PK(3, false, true)}, {OC_PK(3, true, false), OC_PK(3, true, true)}},
                                                     {{OC_PK(4, false, false), OC_PK(4, false, true)}, {OC_PK(4, true, false), OC_PK(4, true, true)}}};
#undef OC_PK
#ifdef OC_DIAGNOSTICS
#define OC_PK(F, C) oc_policy_kernel<3, F, C, false, true, false, true>
    static const kernel_t kernels_fold[2][2] = {{OC_PK(false, false), OC_PK(false, true)}, {OC_PK(true, false), OC_PK(true, true)}};   // (3 digits: the default)
#undef OC_PK
    const bool fold = a.fold_count != nullptr;
    if (fold && (multi || wave_tiles || digits != 3 || s0.row_index))
        return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_policy_forward: the folded env step needs a single-grid launch of 3-digit networks");
#else
    const bool fold = false;                                          // (no FOLD instantiation exists in the product library)
#endif
    const bool gather = !multi && s0.row_index != nullptr;
    kernel_t kernel = wave_tiles ? kernels_wave[digits - 1] : multi ? kernels_multi[digits - 1]
                      : gather ? kernels_gather[digits - 1][fc1_lds ? 1 : 0][conv_lds ? 1 : 0]
                      : kernels[digits - 1][fc1_lds ? 1 : 0][conv_lds ? 1 : 0];
#ifdef OC_DIAGNOSTICS
    if (fold && !wave_tiles && !multi && !gather) kernel = kernels_fold[fc1_lds ? 1 : 0][conv_lds ? 1 : 0];
#endif
    // Round 6: TWO blocks per wavefront (policy_duo_kernel.h) wherever eight row-staging areas fit next to the dense tail -- one
    // wavefront per SIMD interleaving two independent blocks, conv digits in registers, fc1 LDS-resident if that fits as well.  Plain
    // single-grid launches only: gather tiles, select codes, several grids, wave tiles and the diagnostics' fold / forced plans stay
    // on oc_policy_kernel.  OC_POLICY_DUO=0 (read per call) keeps every launch there, for A/B runs and plan-equality checks.
    bool duo = !multi && !wave_tiles && !gather && !fold && !s0.row_select && !(env_plan && env_plan[0]);
    if (duo) {
        // Which launches (profiles/r06_*policy*.log): in the steady state a pair costs 11.4-12.9 us where the two wavefronts of a SIMD
        // take 13.5 for their two blocks (-5..-8 %: 245 760 rows on 240 workgroups, 32 blocks each, 52.5 us against 57.2), but (i) a
        // wavefront's FIRST pair is slower than eight wavefronts' first blocks (cold code, rows and fc1 fragments from furthest away
        // with half as many wavefronts to hide it: 130 560 rows, 16 blocks per workgroup, 38.1 against 37.1) and (ii) a stretch's last
        // blocks, when fewer than five are left over from the rounds of eight, run as single blocks on wavefronts of their own (~7 us
        // with three SIMDs of the CU idle: 131 070 rows, one block left over, 38.5 against 37.1).  So by default only launches of three
        // rounds or more per workgroup whose stretches are whole rounds or leave five or more blocks over.
        // OC_POLICY_DUO=1 forces it wherever it fits, =0 keeps every launch on oc_policy_kernel (A/B runs, plan-equality checks).
        const char *v = getenv("OC_POLICY_DUO");
        if (v && v[0] == '0') duo = false;
        else if (!(v && v[0] == '1')) {
            const long long blocks = (s0.n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
            const long long wgs = a.n_tiles > 0 ? a.n_tiles : 1, per = blocks / wgs, over = blocks - per * wgs;
            const bool planned = a.n_tiles >= 128;                        // (what launch_policy below takes as one tile per workgroup, or close to it)
            // every workgroup's stretch (per or per + 1 blocks) is whole rounds of eight or leaves five or more over
            auto fits = [](long long t) { return t % 8 == 0 || t % 8 >= 5; };
            duo = planned && per >= 24 && fits(per) && (over == 0 || fits(per + 1));
        }
    }
    bool duo_fc1 = false;
    if (duo) {
        duo_fc1 = kb(w_tail_all + 32) + 8 * obs1 <= kLds;
        if (kb((duo_fc1 ? w_tail_all : w_tail_small) + 32) + 8 * obs1 > kLds) duo = false;
    }
    size_t lds_duo = 0;
    int duo_grid = 0;                                                 // 0: the generic duo build, 1 / 2: the builds for the 5x4 / 5x5 grid
    if (duo) {
        static const kernel_t kernels_duo[4][2] = {{oc_policy_duo_kernel<1, false>, oc_policy_duo_kernel<1, true>}, {oc_policy_duo_kernel<2, false>, oc_policy_duo_kernel<2, true>},
                                                   {oc_policy_duo_kernel<3, false>, oc_policy_duo_kernel<3, true>}, {oc_policy_duo_kernel<4, false>, oc_policy_duo_kernel<4, true>}};
        kernel = kernels_duo[digits - 1][duo_fc1 ? 1 : 0];
        // builds for ONE grid (offsets known at compile time, position loop unrolled): the classic layouts' small grids, three digits
        if (digits == 3 && !duo_fc1 && s0.L.W == 5 && s0.L.H == 4) { kernel = oc_policy_duo_kernel<3, false, 5, 4>; duo_grid = 1; }
        if (digits == 3 && !duo_fc1 && s0.L.W == 5 && s0.L.H == 5) { kernel = oc_policy_duo_kernel<3, false, 5, 5>; duo_grid = 2; }
        waves = 4;
        fc1_lds = duo_fc1;
        conv_lds = false;
        a.lds_weight_bytes = static_cast<int>(kb((duo_fc1 ? w_tail_all : w_tail_small) + 32));
        lds_duo = static_cast<size_t>(a.lds_weight_bytes) + 8u * static_cast<size_t>(a.lds_wave_bytes);
    }
    // the GPU that owns the buffers, not whatever device happens to be current: its CU count, its LDS opt-in, its stream
    int dev = oc_device_of_pointer(s0.packed);
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return oc_internal_fail(OC_ERR_HIP, "oc_policy_forward: no current device");
    if (dev < 0 || dev >= 64) return oc_internal_fail(OC_ERR_HIP, "oc_policy_forward: device %d out of range", dev);
    const OcDeviceScope on_device(dev);
    if (!on_device.ok) return oc_internal_fail(OC_ERR_HIP, "oc_policy_forward: cannot make device %d current", dev);
    static size_t lds_granted[64][4][2][2][8] = {};                   // largest opt-in already set per device and kernel instance
    size_t &granted = lds_granted[dev][digits - 1][fc1_lds ? 1 : 0][conv_lds ? 1 : 0][duo ? 5 + duo_grid : wave_tiles ? 2 : multi ? 1 : gather ? 3 : fold ? 4 : 0];
    const size_t lds_use = duo ? lds_duo : lds;
    if (lds_use > 64 * 1024 && lds_use > granted) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds_use));
        if (e != hipSuccess) return oc_internal_fail(OC_ERR_HIP, "oc_policy_forward: %zu bytes of LDS refused: %s", lds_use, hipGetErrorString(e));
        granted = lds_use;
    }
    // one workgroup per CU; each takes an equal share of the 32-row blocks (workgroups without a share return at once)
    static int cus_of[64] = {};
    if (cus_of[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
            return oc_internal_fail(OC_ERR_HIP, "oc_policy_forward: cannot query the CU count");
        cus_of[dev] = v;
    }
    // the launch is sized for the CUs its stream may use: the whole device, or a CU-masked stream's share (oc_exp_cu_partition_create)
    const int n_cus = cu_budget > 0 ? std::min<int>(cu_budget, cus_of[dev]) : cus_of[dev];
    // A table of at least half as many tiles as CUs is taken as planned (oc_policy_plan_tiles balances one per CU):
    // workgroup i runs tile i.  A shorter table (e.g. one entry per segment) is split evenly, block by block.
    // (more tiles than CUs -- many segments on a small CU share -- would need a second round of workgroups: split evenly instead)
    // Several grids: workgroup i owns tile_per_wg consecutive entries of the table (oc_policy_plan_tiles_multi).
    // (a small launch -- at least as many tiles as the rows have blocks, e.g. the 32 one-block tiles the planner cuts 1000 rows into -- is
    // mapped the same way: 32 workgroups instead of 256 that each scan the table to find out they have nothing to do)
    if (!multi) a.tile_per_wg = (a.n_tiles <= n_cus && (a.n_tiles >= n_cus / 2 || (s0.n_rows + kRowsPerBlock - 1) / kRowsPerBlock <= a.n_tiles)) ? 1 : 0;
    static const char *const env_desync = OC_DIAG_ENV("OC_POLICY_DESYNC");
    a.desync = env_desync ? atoi(env_desync) : 0;
    const char *const env_coop = getenv("OC_POLICY_COOP");           // diagnostics (read per call): 0 = every block on one wavefront
    a.coop = env_coop ? atoi(env_coop) != 0 : 1;
    a.coop1 = env_coop && atoi(env_coop) == 4 ? 0 : 1;               // (OC_POLICY_COOP=4: a one-block stretch stays on one wavefront -- A/B runs)
    a.coop2 = env_coop && atoi(env_coop) == 2 ? 2 : 0;               // (OC_POLICY_COOP=2: two leftover blocks of an eight-wavefront stretch shared by two pairs -- A/B runs; measured 1 % slower)
    const char *const env_mono = OC_DIAG_ENV("OC_POLICY_MONO");      // (diagnostics build, read per call) 0 = the plain position loop in the one-wavefront-per-SIMD builds, 2 = timing only
    a.mono = env_mono ? atoi(env_mono) : 1;
    int grid = a.tile_per_wg ? a.n_tiles / a.tile_per_wg : n_cus;
    static const char *const env_grid = OC_DIAG_ENV("OC_POLICY_GRID");
    if (env_grid) { const int g = atoi(env_grid); if (g >= 1 && g <= 65535 && !a.tile_per_wg) grid = g; }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(grid)), dim3(64 * waves), lds_use, stream, a.tiles, a.tile_per_wg, a);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return oc_internal_fail(OC_ERR_HIP, "oc_policy_forward: launch failed: %s", hipGetErrorString(err));
    return OC_OK;
}

}  // namespace

extern "C" {

// While one of these lives, the engine's steps into fresh (time-major) slots store their planes plainly: the launch that follows
// reads them (oc_internal_planes_read_next).
struct PlanesReadNext {
    oc_engine *e;
    oc_group *g;
    explicit PlanesReadNext(oc_engine *e_) : e(e_), g(nullptr) { oc_internal_planes_read_next(e, 1); }
    explicit PlanesReadNext(oc_group *g_) : e(nullptr), g(g_) { set(1); }
    ~PlanesReadNext() { if (g) set(0); else oc_internal_planes_read_next(e, 0); }
    void set(int on) { for (int i = 0; oc_internal_group_engine(g, i); ++i) oc_internal_planes_read_next(oc_internal_group_engine(g, i), on); }
    PlanesReadNext(const PlanesReadNext &) = delete;
    PlanesReadNext &operator=(const PlanesReadNext &) = delete;
};

// the env step folded into the launch (oc_policy_rollout, diagnostics build): the engine's step arguments for this step and the fold groups' counters
#ifdef OC_DIAGNOSTICS
struct FoldParams {
    const oc_dev::StepArgs *args;       // device template (oc_internal_fold_workspace)
    const void *actions;
    int8_t *obs;
    int32_t *reward, *done;
    long long kitchens;
    int32_t *count;
    const int32_t *expected;
    int groups, nt;
    size_t lds_need;
};
#else
struct FoldParams;                      // (never instantiated in the product library)
#endif

static int policy_forward_impl(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                               const int32_t *d_tiles, int32_t n_tiles, const int32_t *d_row_index, const int32_t *d_row_count, int64_t gather_len,
                               float *d_raw, int32_t *d_actions, float *d_action_logp, float *d_logp, int sample_mode, uint32_t seed,
                               uint32_t counter, const uint32_t *d_counter, const int32_t *d_row_select, int32_t cu_budget, void *stream,
                               const FoldParams *fold = nullptr) {
    if (cu_budget < 0) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: negative cu_budget");
    if (!d_tiles || n_tiles < 0 || (reinterpret_cast<uintptr_t>(d_tiles) & 15))
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: d_tiles NULL or not 16-byte aligned");
    if (sample_mode != 0 && sample_mode != 1) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward: sample_mode %d", sample_mode);
    PolicyArgs a;
    memset(&a, 0, sizeof(a));
    if (int rc = fill_slot(a.slot[0], shape, d_packed, n_nets, d_obs, n_rows, d_raw, d_actions, d_action_logp, d_logp, d_row_select, seed)) return rc;
    if (d_row_index || d_row_count)
        if (int rc = fill_gather(a.slot[0], d_row_index, d_row_count, gather_len)) return rc;
    if (n_tiles == 0) return OC_OK;
    a.n_slots = 1;
    a.tiles = reinterpret_cast<const int4 *>(d_tiles);
    a.n_tiles = n_tiles;
    a.counter = counter; a.counter_ptr = d_counter; a.sample = sample_mode;
#ifdef OC_DIAGNOSTICS
    if (fold) {
        a.fold_args = fold->args; a.fold_actions = fold->actions; a.fold_obs = fold->obs; a.fold_reward = fold->reward; a.fold_done = fold->done;
        a.fold_kitchens = fold->kitchens;
        a.fold_count = fold->count; a.fold_expected = fold->expected; a.fold_groups = fold->groups; a.fold_nt = fold->nt;
        if (static_cast<size_t>(staging_bytes(a.slot[0].RB)) < fold->lds_need)
            return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_policy_forward: the env step's LDS block does not fit a row-staging area");
    }
#else
    (void)fold;
#endif
    return launch_policy(a, shape->digits, false, cu_budget, static_cast<hipStream_t>(stream));
}

// (cu_budget > 0: the launch is sized for a CU-masked stream's share -- the CU-partition experiment of round 2, reachable through
// oc_exp_policy_forward_cus in the diagnostics library only; the product's entry points pass 0 = the whole device)
static int policy_forward_cus(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                              const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                              float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                              const int32_t *d_row_select, int32_t cu_budget, void *stream) {
    return policy_forward_impl(shape, d_packed, n_nets, d_obs, n_rows, d_tiles, n_tiles, nullptr, nullptr, 0, d_raw, d_actions, d_action_logp, d_logp,
                               sample_mode, seed, counter, d_counter, d_row_select, cu_budget, stream);
}

int oc_policy_forward_gather(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                             const int32_t *d_tiles, int32_t n_tiles, const int32_t *d_row_index, const int32_t *d_row_count, int64_t gather_len,
                             float *d_raw, int32_t *d_actions, float *d_action_logp, float *d_logp, int sample_mode, uint32_t seed,
                             uint32_t counter, const uint32_t *d_counter, void *stream) {
    if (!d_row_index || !d_row_count) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_gather: NULL index / count arrays");
    return policy_forward_impl(shape, d_packed, n_nets, d_obs, n_rows, d_tiles, n_tiles, d_row_index, d_row_count, gather_len, d_raw, d_actions,
                               d_action_logp, d_logp, sample_mode, seed, counter, d_counter, nullptr, 0, stream);
}

int oc_policy_forward_multi(const oc_policy_multi_slot *slots, int32_t n_slots, const int32_t *d_tiles, int32_t n_tiles, int32_t tiles_per_workgroup,
                            int32_t waves_per_workgroup, int sample_mode, uint32_t counter, const uint32_t *d_counter, void *stream) {
    if (!slots || n_slots < 1 || n_slots > kMaxSlots)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_multi: 1..%d slots", kMaxSlots);
    if (!d_tiles || n_tiles < 0 || (reinterpret_cast<uintptr_t>(d_tiles) & 15) || tiles_per_workgroup < 1 || n_tiles % tiles_per_workgroup)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_multi: d_tiles NULL / not 16-byte aligned, or n_tiles not a multiple of tiles_per_workgroup");
    if (sample_mode != 0 && sample_mode != 1) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_multi: sample_mode %d", sample_mode);
    PolicyArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n_slots; ++i) {
        const oc_policy_multi_slot &m = slots[i];
        if (int rc = fill_slot(a.slot[i], m.shape, m.d_packed, m.n_nets, m.d_obs, m.n_rows, m.d_raw, m.d_actions, m.d_action_logp, m.d_logp,
                               m.d_row_select, m.seed))
            return rc;
        if (m.shape->digits != slots[0].shape->digits)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_multi: every slot's networks must use the same number of conv digits");
        if (oc_device_of_pointer(m.d_packed) != oc_device_of_pointer(slots[0].d_packed))
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_multi: slot %d lives on another device", i);
    }
    if (n_tiles == 0) return OC_OK;
    a.n_slots = n_slots;
    a.tiles = reinterpret_cast<const int4 *>(d_tiles);
    a.n_tiles = n_tiles;
    a.counter = counter; a.counter_ptr = d_counter; a.sample = sample_mode;
    a.tile_per_wg = tiles_per_workgroup;
    if (waves_per_workgroup != 1 && waves_per_workgroup != 4)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_forward_multi: waves_per_workgroup is 4, or 1 for wave tiles (as oc_policy_plan_tiles_multi returned it)");
    return launch_policy(a, slots[0].shape->digits, true, 0, static_cast<hipStream_t>(stream), waves_per_workgroup == 1);
}

int oc_policy_rollout(oc_engine *e, const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int32_t *d_tiles,
                      int32_t n_tiles, int64_t T, int in_place, int8_t *d_obs, int32_t *d_actions, float *d_action_logp,
                      float *d_values, int32_t *d_reward, int32_t *d_done, void *d_records, int sample_mode, uint32_t seed, uint32_t counter0,
                      const int32_t *d_row_select, void *stream) {
    if (!e || !d_obs || !d_actions || !d_reward || !d_done || T < 0)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_rollout: NULL pointer or negative T");
    int64_t shp[5];
    if (int rc = oc_obs_shape(e, shp)) return rc;
    if (shp[0] != 2 || shp[4] != OC_POLICY_C) return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_policy_rollout: the fused policy kernel reads 2-player observations");
    if (!shape || shp[2] != shape->width || shp[3] != shape->height)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_rollout: the networks' grid does not match the engine's");
    const int64_t n = shp[1], rows = 2 * n, slot = rows * shp[2] * shp[3] * shp[4], rec_slot = n * oc_state_record_bytes(e);
    const OcDeviceScope on_device(oc_device_id(e));       // once for the whole rollout; the per-step calls then find it current
    if (!on_device.ok) return oc_internal_fail(OC_ERR_HIP, "oc_policy_rollout: cannot make device %d current", oc_device_id(e));
    // Two launches per step.  (ONE launch per step -- the env step folded into the policy launch, see fold_* above -- was built, is
    // bit-exact and measured 2x slower, DESIGN.md 9b: the diagnostics build keeps it behind OC_POLICY_FOLD=1 for tools/exp_fold_ab.py.)
    bool fold = false;
#ifdef OC_DIAGNOSTICS
    FoldParams fp;
    {
        const char *env_fold = OC_DIAG_ENV("OC_POLICY_FOLD");
        const bool want = env_fold ? (env_fold[0] && env_fold[0] != '0') : kFoldByDefault;
        int32_t *cnt = nullptr, *expd = nullptr, groups = 0;
        const oc_dev::StepArgs *tmpl = nullptr;
        oc_dev::StepArgs probe;
        // (not in place with critics: a group is stepped as soon as its ACTOR blocks have arrived, and a critic block that comes later
        // in the launch would read the rows of step t + 1 where they were overwritten)
        if (want && T > 0 && shape->digits == 3 && !d_row_select && !d_records && n_tiles > 0 && !(in_place && d_values) &&
            oc_internal_fold_args(e, d_actions, d_obs, d_reward, d_done, &probe, &fp.lds_need) == OC_OK &&
            fp.lds_need <= static_cast<size_t>(staging_bytes(static_cast<int>(shp[2] * shp[3] * shp[4]))) &&
            oc_internal_fold_workspace(e, &cnt, &expd, &groups, &tmpl) == OC_OK) {
            hipStream_t st = static_cast<hipStream_t>(stream);
            // the groups' expected arrivals for THIS tile table (cheap: once per rollout call, on the stream)
            if (hipMemsetAsync(expd, 0, static_cast<size_t>(groups) * sizeof(int32_t), st) != hipSuccess)
                return oc_internal_fail(OC_ERR_HIP, "oc_policy_rollout: cannot clear the fold counters");
            hipLaunchKernelGGL(fold_expected_kernel, dim3(static_cast<unsigned>((n_tiles + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const int4 *>(d_tiles), n_tiles, static_cast<long long>(rows), static_cast<long long>(n), expd);
            if (hipGetLastError() != hipSuccess) return oc_internal_fail(OC_ERR_HIP, "oc_policy_rollout: fold_expected_kernel launch failed");
            fp.args = tmpl; fp.kitchens = n;
            fp.count = cnt; fp.expected = expd; fp.groups = groups; fp.nt = in_place ? 0 : 1;
            if (const char *dbg = OC_DIAG_ENV("OC_FOLD_DEBUG")) fp.nt |= atoi(dbg) & ~1;      // timing diagnostics only (results then wrong by design)
            fold = true;
        }
    }
#endif
    const PlanesReadNext hint(e);                                     // (every step's planes are the next policy launch's input)
    for (int64_t t = 0; t < T; ++t) {
        const int64_t k = in_place ? 0 : t;
        int8_t *obs_t = d_obs + k * slot, *obs_next = in_place ? d_obs : d_obs + (t + 1) * slot;
        int32_t *act_t = d_actions + k * rows;
#ifdef OC_DIAGNOSTICS
        if (fold) {
            fp.actions = act_t; fp.obs = obs_next; fp.reward = d_reward + k * rows; fp.done = d_done + k * n;
            if (int rc = policy_forward_impl(shape, d_packed, n_nets, obs_t, rows, d_tiles, n_tiles, nullptr, nullptr, 0, d_values ? d_values + k * rows : nullptr,
                                             act_t, d_action_logp ? d_action_logp + k * rows : nullptr, nullptr, sample_mode, seed,
                                             counter0 + static_cast<uint32_t>(t), nullptr, nullptr, 0, stream, &fp))
                return rc;
            oc_internal_note_destination(e, obs_next);
            ++g_fold_launches;
            continue;
        }
#endif
        if (int rc = oc_policy_forward(shape, d_packed, n_nets, obs_t, rows, d_tiles, n_tiles, d_values ? d_values + k * rows : nullptr, act_t,
                                       d_action_logp ? d_action_logp + k * rows : nullptr, nullptr, sample_mode, seed,
                                       counter0 + static_cast<uint32_t>(t), nullptr, d_row_select ? d_row_select + t * rows : nullptr, stream))
            return rc;
        if (int rc = oc_step(e, act_t, OC_ACT_I32, obs_next, d_reward + k * rows, d_done + k * n,
                             d_records ? static_cast<char *>(d_records) + (t + 1) * rec_slot : nullptr, stream)) return rc;
    }
    return OC_OK;
}

int oc_policy_rollout_mixed(oc_engine *e, const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int32_t *d_tiles,
                            int32_t n_tiles, int64_t T, int8_t *d_obs, int32_t *d_actions, float *d_action_logp, float *d_values,
                            int32_t *d_reward, int32_t *d_done, void *d_records, int sample_mode, uint32_t seed, uint32_t counter0,
                            const int32_t *d_choice, int32_t *d_index_ws, int32_t *d_count_ws, void *stream) {
    if (!e || !d_obs || !d_actions || !d_reward || !d_done || !d_choice || !d_index_ws || !d_count_ws || T < 0)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_rollout_mixed: NULL pointer or negative T");
    int64_t shp[5];
    if (int rc = oc_obs_shape(e, shp)) return rc;
    if (shp[0] != 2 || shp[4] != OC_POLICY_C) return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_policy_rollout_mixed: the fused policy kernel reads 2-player observations");
    if (!shape || shp[2] != shape->width || shp[3] != shape->height)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_policy_rollout_mixed: the networks' grid does not match the engine's");
    const int64_t n = shp[1], rows = 2 * n, slot = rows * shp[2] * shp[3] * shp[4], rec_slot = n * oc_state_record_bytes(e);
    const OcDeviceScope on_device(oc_device_id(e));
    if (!on_device.ok) return oc_internal_fail(OC_ERR_HIP, "oc_policy_rollout_mixed: cannot make device %d current", oc_device_id(e));
    // every step's rows sorted by who plays them, for all steps at once: list 2t + side holds row ids side * N + kitchen
    if (int rc = oc_partition_rows(d_choice, 2 * T, n, 2, d_index_ws, d_count_ws, stream)) return rc;
    const PlanesReadNext hint(e);
    for (int64_t t = 0; t < T; ++t) {
        int32_t *act_t = d_actions + t * rows;
        // (step t reads slot t and the env step writes slot t + 1: the 12 bytes a ragged row's last piece may over-read lie inside the buffer)
        if (int rc = oc_policy_forward_gather(shape, d_packed, n_nets, d_obs + t * slot, rows, d_tiles, n_tiles, d_index_ws + t * rows, d_count_ws + t * 4, n,
                                              d_values ? d_values + t * rows : nullptr, act_t, d_action_logp ? d_action_logp + t * rows : nullptr,
                                              nullptr, sample_mode, seed, counter0 + static_cast<uint32_t>(t), nullptr, stream))
            return rc;
        if (int rc = oc_step(e, act_t, OC_ACT_I32, d_obs + (t + 1) * slot, d_reward + t * rows, d_done + t * n,
                             d_records ? static_cast<char *>(d_records) + (t + 1) * rec_slot : nullptr, stream)) return rc;
    }
    return OC_OK;
}

#ifdef OC_DIAGNOSTICS
int oc_exp_cu_partition_create(int device_id, const float *shares, int32_t n, void **out_streams, int32_t *out_cus) {
    if (!shares || !out_streams || n < 1 || n > 64) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_exp_cu_partition_create: 1..64 streams");
    const OcDeviceScope on_device(device_id);
    if (!on_device.ok) return oc_internal_fail(OC_ERR_HIP, "oc_exp_cu_partition_create: cannot make device %d current", device_id);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus < n || cus > 1024)
        return oc_internal_fail(OC_ERR_HIP, "oc_exp_cu_partition_create: %d CUs cannot be shared out over %d streams", cus, n);
    double total = 0;
    for (int i = 0; i < n; ++i) {
        if (!(shares[i] > 0.f)) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_exp_cu_partition_create: shares must be positive");
        total += shares[i];
    }
    // CUs per stream: proportional, at least one, largest remainders first
    std::vector<int> want(n);
    int used = 0;
    for (int i = 0; i < n; ++i) { want[i] = std::max(1, static_cast<int>(shares[i] / total * cus)); used += want[i]; }
    while (used < cus) {
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (shares[i] / want[i] > shares[best] / want[best]) best = i;
        ++want[best]; ++used;
    }
    while (used > cus) {
        int best = -1;
        for (int i = 0; i < n; ++i)
            if (want[i] > 1 && (best < 0 || shares[i] / want[i] < shares[best] / want[best])) best = i;
        --want[best]; --used;
    }
    // every stream's CUs spread over the whole index range (weighted round robin), so each share touches every XCD / memory channel
    std::vector<std::vector<uint32_t>> mask(n, std::vector<uint32_t>((cus + 31) / 32, 0u));
    std::vector<int> given(n, 0);
    for (int c = 0; c < cus; ++c) {
        int best = -1;
        double lag = -1e30;
        for (int i = 0; i < n; ++i) {
            if (given[i] >= want[i]) continue;
            const double l = static_cast<double>(c + 1) * want[i] / cus - given[i];     // how far behind its even pace stream i is
            if (l > lag) { lag = l; best = i; }
        }
        mask[best][c / 32] |= 1u << (c % 32);
        ++given[best];
    }
    for (int i = 0; i < n; ++i) out_streams[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        hipStream_t st = nullptr;
        const hipError_t e = hipExtStreamCreateWithCUMask(&st, static_cast<uint32_t>(mask[i].size()), mask[i].data());
        if (e != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipStreamDestroy(static_cast<hipStream_t>(out_streams[j]));
            return oc_internal_fail(OC_ERR_HIP, "oc_exp_cu_partition_create: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
        }
        out_streams[i] = st;
        if (out_cus) out_cus[i] = want[i];
    }
    return OC_OK;
}

int oc_exp_cu_partition_destroy(void *const *streams, int32_t n) {
    if (!streams) return OC_OK;
    for (int i = 0; i < n; ++i)
        if (streams[i]) {
            (void)hipStreamSynchronize(static_cast<hipStream_t>(streams[i]));
            (void)hipStreamDestroy(static_cast<hipStream_t>(streams[i]));
        }
    return OC_OK;
}

int oc_exp_policy_forward_cus(const oc_policy_shape *shape, const void *d_packed, int32_t n_nets, const int8_t *d_obs, int64_t n_rows,
                              const int32_t *d_tiles, int32_t n_tiles, float *d_raw, int32_t *d_actions, float *d_action_logp,
                              float *d_logp, int sample_mode, uint32_t seed, uint32_t counter, const uint32_t *d_counter,
                              const int32_t *d_row_select, int32_t cu_budget, void *stream) {
    return policy_forward_cus(shape, d_packed, n_nets, d_obs, n_rows, d_tiles, n_tiles, d_raw, d_actions, d_action_logp, d_logp, sample_mode, seed, counter,
                              d_counter, d_row_select, cu_budget, stream);
}
int oc_exp_group_policy_rollout_streams(oc_group *g, const oc_group_policy_slot *slots, void *const *policy_streams, const int32_t *cu_budgets,
                                        int64_t T, int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream) {
    return group_policy_rollout_streams(g, slots, policy_streams, cu_budgets, T, in_place, sample_mode, seed, counter0, stream);
}
#endif  // OC_DIAGNOSTICS: the oc_exp_* entry points

int oc_group_policy_rollout(oc_group *g, const oc_group_policy_slot *slots, int64_t T, int in_place, int sample_mode, uint32_t seed,
                            uint32_t counter0, void *stream) {
    return group_policy_rollout_streams(g, slots, nullptr, nullptr, T, in_place, sample_mode, seed, counter0, stream);
}

static int group_policy_rollout_streams(oc_group *g, const oc_group_policy_slot *slots, void *const *policy_streams, const int32_t *cu_budgets,
                                        int64_t T, int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream) {
    const int n = oc_group_size(g);
    if (n < 1 || !slots || T < 0) return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_group_policy_rollout: NULL group / slots or negative T");
    std::vector<oc_engine *> eng(n);
    std::vector<int64_t> kitchens(n), rows(n), slot_bytes(n), rec_bytes(n);
    for (int i = 0; i < n; ++i) {
        eng[i] = oc_internal_group_engine(g, i);
        int64_t shp[5];
        if (int rc = oc_obs_shape(eng[i], shp)) return rc;
        const oc_group_policy_slot &s = slots[i];
        if (!s.shape || !s.d_packed || !s.d_tiles || !s.d_obs || !s.d_actions || !s.d_reward || !s.d_done)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_group_policy_rollout: NULL pointer in slot %d", i);
        // (a group may hold engines with 1, 3 or 4 players: their observations are not what the policy kernel reads, and their
        // action / reward buffers are P x N, not 2 x N)
        if (shp[0] != 2 || shp[4] != OC_POLICY_C)
            return oc_internal_fail(OC_ERR_UNSUPPORTED, "oc_group_policy_rollout: engine %d has %lld players; the fused policy kernel reads 2-player observations", i, (long long)shp[0]);
        if (shp[2] != s.shape->width || shp[3] != s.shape->height)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_group_policy_rollout: the networks' grid of slot %d does not match its engine's", i);
        kitchens[i] = shp[1]; rows[i] = 2 * shp[1]; slot_bytes[i] = rows[i] * shp[2] * shp[3] * shp[4];
        rec_bytes[i] = shp[1] * oc_state_record_bytes(eng[i]);
    }
    const OcDeviceScope on_device(oc_device_id(eng[0]));
    if (!on_device.ok) return oc_internal_fail(OC_ERR_HIP, "oc_group_policy_rollout: cannot make device %d current", oc_device_id(eng[0]));
    hipStream_t main_stream = static_cast<hipStream_t>(stream);
    const PlanesReadNext hint(g);                                     // (every step's planes are the next policy launches' input)
    bool side = policy_streams != nullptr;
    for (int i = 0; i < n && side; ++i) side = policy_streams[i] != nullptr && policy_streams[i] != stream;
    auto policy = [&](int i, int64_t t, hipStream_t st, int budget) {
        const oc_group_policy_slot &s = slots[i];
        const int64_t k = in_place ? 0 : t;
        return policy_forward_cus(s.shape, s.d_packed, s.n_nets, s.d_obs + k * slot_bytes[i], rows[i], s.d_tiles, s.n_tiles,
                                     s.d_values ? s.d_values + k * rows[i] : nullptr, s.d_actions + k * rows[i],
                                     s.d_action_logp ? s.d_action_logp + k * rows[i] : nullptr, nullptr, sample_mode,
                                     seed + static_cast<uint32_t>(i) * 0x9E3779B9u, counter0 + static_cast<uint32_t>(t), nullptr, nullptr, budget, st);
    };
    if (!side) {
        // one stream: the engines' policy launches one after the other, then ONE env launch for all of them
        std::vector<const void *> act(n);
        std::vector<int8_t *> obs_next(n);
        std::vector<int32_t *> rew(n), done(n);
        std::vector<void *> recs(n);
        for (int64_t t = 0; t < T; ++t) {
            const int64_t k = in_place ? 0 : t;
            for (int i = 0; i < n; ++i) {
                if (int rc = policy(i, t, main_stream, 0)) return rc;
                act[i] = slots[i].d_actions + k * rows[i];
                obs_next[i] = in_place ? slots[i].d_obs : slots[i].d_obs + (t + 1) * slot_bytes[i];
                rew[i] = slots[i].d_reward + k * rows[i];
                done[i] = slots[i].d_done + k * kitchens[i];
                recs[i] = slots[i].d_records ? static_cast<char *>(slots[i].d_records) + (t + 1) * rec_bytes[i] : nullptr;
            }
            if (int rc = oc_group_step(g, act.data(), OC_ACT_I32, obs_next.data(), rew.data(), done.data(), recs.data(), stream)) return rc;
        }
        return OC_OK;
    }
    // CU-partitioned: kitchens of different engines never interact, so every engine runs its whole rollout -- policy launch and
    // its own env launch per step -- on its own stream and its own CUs, with no synchronisation between engines at all: one fork
    // off the caller's stream at the start, one join at the end.  (Joining the five streams at every step for a common env launch
    // was measured first: 1.3 ms per joint step, the cross-queue dependencies cost ten times the kernels.)
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    auto cleanup = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    hipError_t he = hipSuccess;
    for (int i = 0; i <= n && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    if (he == hipSuccess) he = hipEventRecord(ev[n], main_stream);
    for (int i = 0; i < n && he == hipSuccess; ++i) he = hipStreamWaitEvent(static_cast<hipStream_t>(policy_streams[i]), ev[n], 0);
    if (he != hipSuccess) { cleanup(); return oc_internal_fail(OC_ERR_HIP, "oc_group_policy_rollout: fork failed: %s", hipGetErrorString(he)); }
    int rc = OC_OK;
    for (int i = 0; i < n; ++i) (void)oc_internal_set_cu_budget(eng[i], cu_budgets ? cu_budgets[i] : 0);
    for (int64_t t = 0; t < T && rc == OC_OK; ++t) {
        const int64_t k = in_place ? 0 : t;
        for (int i = 0; i < n && rc == OC_OK; ++i) {                     // interleaved, so that every stream always has work queued
            const oc_group_policy_slot &s = slots[i];
            rc = policy(i, t, static_cast<hipStream_t>(policy_streams[i]), cu_budgets ? cu_budgets[i] : 0);
            if (rc == OC_OK)
                rc = oc_step(eng[i], s.d_actions + k * rows[i], OC_ACT_I32, in_place ? s.d_obs : s.d_obs + (t + 1) * slot_bytes[i],
                             s.d_reward + k * rows[i], s.d_done + k * kitchens[i],
                             s.d_records ? static_cast<char *>(s.d_records) + (t + 1) * rec_bytes[i] : nullptr, policy_streams[i]);
        }
    }
    for (int i = 0; i < n; ++i) (void)oc_internal_set_cu_budget(eng[i], 0);
    for (int i = 0; i < n && he == hipSuccess; ++i) {
        he = hipEventRecord(ev[i], static_cast<hipStream_t>(policy_streams[i]));
        if (he == hipSuccess) he = hipStreamWaitEvent(main_stream, ev[i], 0);
    }
    cleanup();
    if (rc == OC_OK && he != hipSuccess) rc = oc_internal_fail(OC_ERR_HIP, "oc_group_policy_rollout: join failed: %s", hipGetErrorString(he));
    return rc;
}

int oc_group_policy_rollout_multi(oc_group *g, const oc_group_policy_slot *slots, const int32_t *d_tiles, int32_t n_tiles, int32_t tiles_per_workgroup,
                                  int32_t waves_per_workgroup, int64_t T, int in_place, int sample_mode, uint32_t seed, uint32_t counter0, void *stream) {
    const int n = oc_group_size(g);
    if (n < 1 || n > kMaxSlots || !slots || !d_tiles || T < 0)
        return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_group_policy_rollout_multi: NULL group / slots / tiles, negative T or more than %d engines", kMaxSlots);
    std::vector<int64_t> kitchens(n), rows(n), slot_bytes(n), rec_bytes(n);
    for (int i = 0; i < n; ++i) {
        int64_t shp[5];
        if (int rc = oc_obs_shape(oc_internal_group_engine(g, i), shp)) return rc;
        rec_bytes[i] = shp[1] * oc_state_record_bytes(oc_internal_group_engine(g, i));
        const oc_group_policy_slot &s = slots[i];
        if (!s.shape || !s.d_packed || !s.d_obs || !s.d_actions || !s.d_reward || !s.d_done)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_group_policy_rollout_multi: NULL pointer in slot %d", i);
        if (shp[0] != 2 || shp[2] != s.shape->width || shp[3] != s.shape->height)
            return oc_internal_fail(OC_ERR_INVALID_ARG, "oc_group_policy_rollout_multi: the networks' grid of slot %d does not match its (2-player) engine's", i);
        kitchens[i] = shp[1]; rows[i] = 2 * shp[1]; slot_bytes[i] = rows[i] * shp[2] * shp[3] * shp[4];
    }
    const OcDeviceScope on_device(oc_device_id(oc_internal_group_engine(g, 0)));
    if (!on_device.ok) return oc_internal_fail(OC_ERR_HIP, "oc_group_policy_rollout_multi: cannot make the group's device current");
    std::vector<oc_policy_multi_slot> ms(n);
    std::vector<const void *> act(n);
    std::vector<int8_t *> obs_next(n);
    std::vector<int32_t *> rew(n), done(n);
    std::vector<void *> recs(n);
    const PlanesReadNext hint(g);
    for (int64_t t = 0; t < T; ++t) {
        const int64_t k = in_place ? 0 : t;
        for (int i = 0; i < n; ++i) {
            const oc_group_policy_slot &s = slots[i];
            oc_policy_multi_slot &m = ms[i];
            m.shape = s.shape; m.d_packed = s.d_packed; m.n_nets = s.n_nets;
            m.d_obs = s.d_obs + k * slot_bytes[i]; m.n_rows = rows[i];
            m.d_raw = s.d_values ? s.d_values + k * rows[i] : nullptr;
            m.d_actions = s.d_actions + k * rows[i];
            m.d_action_logp = s.d_action_logp ? s.d_action_logp + k * rows[i] : nullptr;
            m.d_logp = nullptr; m.d_row_select = nullptr;
            m.seed = seed + static_cast<uint32_t>(i) * 0x9E3779B9u;      // as oc_group_policy_rollout draws them
            act[i] = m.d_actions;
            obs_next[i] = in_place ? s.d_obs : s.d_obs + (t + 1) * slot_bytes[i];
            rew[i] = s.d_reward + k * rows[i];
            done[i] = s.d_done + k * kitchens[i];
            recs[i] = s.d_records ? static_cast<char *>(s.d_records) + (t + 1) * rec_bytes[i] : nullptr;
        }
        if (int rc = oc_policy_forward_multi(ms.data(), n, d_tiles, n_tiles, tiles_per_workgroup, waves_per_workgroup, sample_mode, counter0 + static_cast<uint32_t>(t), nullptr, stream)) return rc;
        if (int rc = oc_group_step(g, act.data(), OC_ACT_I32, obs_next.data(), rew.data(), done.data(), recs.data(), stream)) return rc;
    }
    return OC_OK;
}

#ifdef OC_DIAGNOSTICS
/* Diagnostics build (overcooked_debug.h): per-wavefront phase stamps of the following launches
 * (tools/trace_policy.py).  enable: allocate and zero; otherwise copy [n_waves][16] stamps out and free. */
int oc_policy_debug_trace(int enable, unsigned long long *host_out, int64_t n_waves) {
    const size_t cap = 4096 * 4 * 32 * sizeof(unsigned long long);
    if (enable) {
        if (!g_trace && hipMalloc(reinterpret_cast<void **>(&g_trace), cap) != hipSuccess) return OC_ERR_HIP;
        return hipMemset(g_trace, 0, cap) == hipSuccess ? OC_OK : OC_ERR_HIP;
    }
    if (!g_trace) return OC_OK;
    (void)hipDeviceSynchronize();
    if (host_out) (void)hipMemcpy(host_out, g_trace, static_cast<size_t>(n_waves < 4096 * 4 ? n_waves : 4096 * 4) * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(g_trace);
    g_trace = nullptr;
    return OC_OK;
}

long long oc_debug_fold_launches(void) { return g_fold_launches; }
#endif  // OC_DIAGNOSTICS

float oc_policy_uniform(uint32_t seed, uint32_t counter, uint32_t row, uint32_t action) {
    return oc_policy_uniform_from_hash(oc_policy_hash(seed, counter, row, action));
}

}  // extern "C"
