// policy_duo_kernel.h -- oc_policy_duo_kernel: the fused policy kernel with TWO 32-row blocks per wavefront (round 6), included by
// policy_ops.hip inside its anonymous namespace (it uses that file's helpers: stage_rows, read_bfrags, mfma_f16, split_pair,
// relu_clamp, copy_to_lds, policy_find_stretch ...).  Same arithmetic, same packed image, same outputs bit for bit as
// oc_policy_kernel's ordinary blocks (policy_block_body.h; restated lane for lane in oracle/policy_oracle.py) -- what changes is who
// issues what when.
//
// Why.  A gfx950 SIMD issues one wavefront's instructions in order; an MFMA occupies the matrix pipe for 32 cycles but the SIMD's
// issue port for 8, so up to ~5 single-issue vector instructions placed BEHIND an MFMA in program order run under it for free
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").  One block's chain is conv (18 integer MFMAs, no vector work of its own)
// -> operand conversion (~130 vector instructions, no matrix work) -> fc1 (12 MFMAs): inside one block there is nothing to put under
// the conv's MFMAs but the previous position's conversion, which the fc1 of that position is waiting for.  Rounds 2-5 ran two
// wavefronts per SIMD and let the hardware arbiter mix them: whenever both are in the same phase the matrix pipe or the vector port
// idles (66 % matrix occupancy in the position loop, 0.32 of the issue bound over the launch).  Here ONE wavefront per
// SIMD owns two independent blocks A and B and interleaves them by construction:
//
//     half-step A(p):   MFMAs  A.conv(p+1)  A.fc1(p)      ||  vector  B.convert(p)       (B's conv(p) finished a half-step ago)
//     half-step B(p):   MFMAs  B.conv(p+1)  B.fc1(p)      ||  vector  A.convert(p+1)     (A's conv(p+1) finished a half-step ago)
//
// 30 MFMAs (960 matrix cycles) beside ~130 vector instructions per half-step: 4-5 per MFMA gap, pinned with sched_group_barrier.
// Conv weight digits (72 registers) and the fc1 fragments of a position (32) are read ONCE for both blocks and live in AccVGPRs (they
// are only ever MFMA A operands); the dense tail (fc2, head) interleaves the two blocks the same way, one block's operand split under
// the other's MFMAs.
//
// Shape.  Workgroup = 4 wavefronts (one per SIMD, up to 512 registers), conv digits in registers, the dense tail (and fc1 when it
// fits) in LDS, two row-staging areas per wavefront.  Needs 8 staging areas + the tail in 160 KB of LDS: every grid of up to 25
// tiles (cramped_room, the 5x5 layouts); larger grids, gather tiles, select codes, several grids per launch and wave tiles stay on
// oc_policy_kernel (launch_policy decides).  A stretch's blocks go to the wavefronts as pairs, eight blocks per round; what is left
// (fewer than eight) runs as pairs and -- on the wavefronts that would otherwise idle -- single blocks through the one-block body.

// The interleave of a half-step, pinned for the scheduler: N times [one matrix instruction, then the vector instructions that run
// under it], an LDS read added from the FIRST_DS-th gap on (masks: 0x008 MFMA, 0x002 VALU, 0x100 LDS read).  The counts are upper
// bounds -- a group takes what is there -- and the builtin wants literal constants, hence the recursion.

template <int N, int FIRST_DS, int I = 0>
__device__ __forceinline__ void oc_pin_interleave() {
    if constexpr (I < N) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (I & 3) == 0 ? 5 : 4, 0);
        if constexpr (I >= FIRST_DS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        oc_pin_interleave<N, FIRST_DS, I + 1>();
    }
}

// GW, GH: the grid this build is for (0, 0: any grid, read from the launch arguments) -- see policy_duo_body.h
template <int NDIG, bool FC1_LDS, int GW = 0, int GH = 0>
__global__ __launch_bounds__(256) void oc_policy_duo_kernel(const int4 *pre_tiles, int pre_tile_per_wg, const PolicyArgs a) {
    // (names and constants the one-block body, included below for single blocks, expects from its kernel)
    constexpr bool CONV_LDS = false, TAIL_LDS = true, FOLD = false;
    constexpr bool gather = false, leader = false, in_coop = false;
    constexpr int grp = 0, gsz = 0, member = 0, coop_lead = 0, coop_seq = 0;
    constexpr bool kCoopFlags = false, kMono = false;                 // (single blocks here run the plain loop of policy_block_body.h)
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds_raw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    // two staging areas per wavefront: block A's (also a single block's) and block B's
    uint8_t *ldsw = lds + a.lds_weight_bytes + static_cast<size_t>(2 * wave) * a.lds_wave_bytes;
    uint8_t *ldsw2 = ldsw + a.lds_wave_bytes;

    long long wg0, wg1, tstart;
    int ti;
    const int own_tile_word = prefetch_own_tile(pre_tiles, pre_tile_per_wg);
    warm_kernarg<false>();
    keep_prefetched(own_tile_word);
    stamp(a, wave, lane, 9);
    if (!policy_find_stretch<false>(a, lane, wg0, wg1, ti, tstart)) return;
    stamp(a, wave, lane, 0); stamp(a, wave, lane, 14);

    const PolicySlot &S = a.slot[0];
    const int RB = S.RB, P = S.L.P, PH = S.L.PH, H = S.L.H;
    const int RS = RB;
    const uint32_t counter = a.counter + (a.counter_ptr ? *a.counter_ptr : 0u);
    const oc_policy_layout &L0 = S.L;
    const int wstart = static_cast<int>(FC1_LDS ? L0.off_fc1 : L0.off_b1);
    uint8_t *lds_tail = lds;                                          // LDS address of image offset wstart (no conv part in LDS)
    v4i wA[6][NDIG];                                                  // conv weight digits [kx*2+s][digit]: MFMA A operands of both blocks
    v16i cbeta;                                                       // conv bias in units of the channel's scale: the lowest digit's accumulator starts from it
    __amdgpu_buffer_rsrc_t w1g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(S.packed), 0, 0, 0x00020000);
    const uint32_t lane16 = static_cast<uint32_t>(lane) * 16u;
    const v4i *w1l = reinterpret_cast<const v4i *>(lds_tail + (L0.off_fc1 - wstart)) + lane;
    const float *b1 = reinterpret_cast<const float *>(lds_tail + (L0.off_b1 - wstart)) + h * 16;
    const v4i *w2 = reinterpret_cast<const v4i *>(lds_tail + (L0.off_fc2 - wstart)) + lane;
    const float *b2 = reinterpret_cast<const float *>(lds_tail + (L0.off_b2 - wstart)) + h * 16;
    const bool head_lane = col < 8;
    const v4i *wh = reinterpret_cast<const v4i *>(lds_tail + (L0.off_head - wstart)) + (8 * h + (col & 7));
    const float *bh = reinterpret_cast<const float *>(lds_tail + (L0.off_bh - wstart)) + h * 4;
    const uint8_t *rowp = ldsw + col * RB + 16 * h;
    const int colstep = H * OC_POLICY_C;
    int loaded_net = -1, n_out = 1;
    float ds1 = 1.f, ds2 = 1.f, ds3 = 1.f;
    bool trace_done = false, trace2_done = false;
    int myrow_next = 0;
    long long fold_r0 = 0, fold_r1 = -1;
    auto fold_drain = [&]() {};
    auto fold_arrive = [&](long long, long long) {};
    (void)fold_r0; (void)fold_r1; (void)myrow_next; (void)leader; (void)in_coop; (void)grp; (void)gsz; (void)member; (void)coop_lead; (void)lds_tail;

    // conv result -> fc1 operand (see oc_policy_kernel: the same arithmetic, but with scalar multiplies -- a packed f32 instruction
    // beside MFMAs costs its issue slot AND ~22 cycles, MI355X_MICROARCH.md)
    auto combine_pair = [&](const v16i (&acc)[NDIG], int r, float &x0, float &x1) {
        constexpr int kB = OC_POLICY_DIGIT_BASE;
        const float sc = oc_pow2f(-oc_policy_conv_shift(NDIG));
        float t0, t1;
        if (NDIG == 1) {
            t0 = static_cast<float>(acc[0][r]) * sc; t1 = static_cast<float>(acc[0][r + 1]) * sc;
        } else {
            t0 = static_cast<float>(acc[0][r] * kB + acc[1][r]); t1 = static_cast<float>(acc[0][r + 1] * kB + acc[1][r + 1]);
            if (NDIG == 2) { t0 = t0 * sc; t1 = t1 * sc; }
            if (NDIG == 3) {
                const float u0 = static_cast<float>(acc[2][r]) * sc, u1 = static_cast<float>(acc[2][r + 1]) * sc;
                t0 = __builtin_fmaf(t0, kB * sc, u0); t1 = __builtin_fmaf(t1, kB * sc, u1);
            }
            if (NDIG == 4) {
                const float u0 = static_cast<float>(acc[2][r] * kB + acc[3][r]) * sc, u1 = static_cast<float>(acc[2][r + 1] * kB + acc[3][r + 1]) * sc;
                t0 = __builtin_fmaf(t0, kB * kB * sc, u0); t1 = __builtin_fmaf(t1, kB * kB * sc, u1);
            }
        }
        x0 = relu_clamp(t0);
        x1 = relu_clamp(t1);
    };
    auto conv = [&](v16i (&acc)[NDIG], const v4i (&bfr)[6]) {
#pragma unroll
        for (int d = 0; d + 1 < NDIG; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][r] = 0;
        acc[NDIG - 1] = cbeta;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks)
#pragma unroll
            for (int d = 0; d < NDIG; ++d)
                acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wA[ks][d], bfr[ks], acc[d], 0, 0, 0);
    };

  for (long long cur = wg0; cur < wg1;) {
    const int4 tile = a.tile_per_wg ? load_tile(a.tiles, ti) : a.tiles[ti];
    const int tnb = policy_tile_blocks<false>(S, tile);
    if (tnb == 0) { ++ti; continue; }
    const long long run_end = wg1 < tstart + tnb ? wg1 : tstart + tnb;
    const int first = static_cast<int>(cur - tstart), last = static_cast<int>(run_end - tstart);   // this stretch of the tile
    const long long row0 = tile.x;
    const int nrows = static_cast<int>(row0 + tile.y > S.n_rows ? S.n_rows - row0 : tile.y);
    const int net = min(max(tile.z, 0), S.n_nets - 1);
    const int flags = tile.w;
    cur = run_end;
    if (run_end == tstart + tnb) { tstart += tnb; ++ti; }

    // ---- this wavefront's items of the stretch: `rounds` pairs (eight blocks per round over the four wavefronts), then at most one
    // more item out of the remaining rem < 8 blocks: rem <= 4 -> single blocks on the first rem wavefronts; else rem - 4 pairs on the
    // first wavefronts and single blocks on the others, so that nobody waits for a wavefront with two blocks while others have none
    const int nb = last - first, rounds = nb / (2 * nwaves), rem = nb - rounds * 2 * nwaves, base = first + rounds * 2 * nwaves;
    const int tail_pairs = rem > nwaves ? rem - nwaves : 0;
    int tailA = -1, tailB = -1;
    if (wave < tail_pairs) { tailA = base + 2 * wave; tailB = tailA + 1; }
    else if (rem > nwaves) { tailA = base + 2 * tail_pairs + (wave - tail_pairs); }
    else if (wave < rem) { tailA = base + wave; }
    const int n_items = rounds + (tailA >= 0 ? 1 : 0);
    auto item_blocks = [&](int i, int &bA, int &bB) {
        if (i < rounds) { bA = first + i * 2 * nwaves + 2 * wave; bB = bA + 1; }
        else { bA = tailA; bB = tailB; }
    };
    auto stage_item = [&](int i) {
        int bA, bB;
        item_blocks(i, bA, bB);
        (void)stage_rows(S, ldsw, row0 + static_cast<long long>(bA) * kRowsPerBlock, lane);
        if (bB >= 0) (void)stage_rows(S, ldsw2, row0 + static_cast<long long>(bB) * kRowsPerBlock, lane);
    };
    auto stage = [&](int) -> int { return -1; };                     // (the one-block body's prefetch hook: single blocks are always a wavefront's last item)
    (void)stage;

    // rows first (they come from furthest away), then the weights
    if (loaded_net < 0) stamp(a, wave, lane, 10);
    if (n_items > 0) stage_item(0);
    const bool reload = net != loaded_net;
    if (reload) {
        if (loaded_net < 0) stamp(a, wave, lane, 11);
        if (loaded_net >= 0) __syncthreads();                         // everybody is done with the previous network
        const uint8_t *blob = S.packed + static_cast<size_t>(net) * S.L.total;
        const int tail_bytes = static_cast<int>(S.L.off_ds + 32 - wstart);
        {
            int wv = wave;
            asm volatile("" : "+s"(wv));
            const int nk = (tail_bytes + 32 + 1023) >> 10;
            copy_to_lds(blob + wstart, lds_tail, nk, (tail_bytes >> 4) - (nk - 1) * 64, blob, wv, nwaves, lane);
        }
        {
            const v4i *src = reinterpret_cast<const v4i *>(blob + S.L.off_conv);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks)
#pragma unroll
                for (int d = 0; d < NDIG; ++d) wA[ks][d] = src[(ks * NDIG + d) * 64 + lane];
            cbeta = *reinterpret_cast<const v16i *>(blob + S.L.off_conv_sb + 128 + h * 64);
        }
        w1g = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(blob + S.L.off_fc1), 0, P * OC_POLICY_FC1_POS_BYTES, 0x00020000);
        if (loaded_net < 0) stamp(a, wave, lane, 12);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wavefront's share of the weight copy (and its rows) have landed
        if (loaded_net < 0) stamp(a, wave, lane, 13);
        asm volatile("s_barrier" ::: "memory");
        n_out = min(__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int *>(lds_tail + tail_bytes + 16)), kMaxOut);   // header word 4, copied behind the tail
        {
            const int *dsp = reinterpret_cast<const int *>(lds_tail + tail_bytes - 32);
            const int d1 = dsp[0], d2 = dsp[1], d3 = dsp[2];
            ds1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(d1));
            ds2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(d2));
            ds3 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(d3));
        }
        loaded_net = net;
        stamp(a, wave, lane, 1);
    }

    for (int item = 0; item < n_items; ++item) {
        int bA, bB;
        item_blocks(item, bA, bB);
        if (bB < 0) {
            // ---- a single block: the one-block body (policy_block_body.h), weights in registers, nothing to prefetch behind it
            const int last_full = 0;
            for (int blk = bA, once = 0; once < 1; ++once) {
#define OC_BLOCK_COOP 0
#include "policy_block_body.h"
#undef OC_BLOCK_COOP
            }
            continue;
        }
        // ---- a pair of blocks, interleaved
#include "policy_duo_body.h"
    }
  }
    stamp(a, wave, lane, 8); stamp(a, wave, lane, 15);
}
