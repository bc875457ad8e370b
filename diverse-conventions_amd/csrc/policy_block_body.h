// policy_block_body.h -- one 32-row block of oc_policy_kernel (policy_ops.hip), included TWICE inside the kernel: as the body of the
// loop over a wavefront's own blocks (OC_BLOCK_COOP 0) and, inside a do { } while (false), as the cooperative block that the
// wavefronts of a group share (OC_BLOCK_COOP 1; see "cooperative blocks" in the kernel).  Two textual instances rather than a runtime
// flag or a generic lambda: with a position stride the compiler cannot see is 1 the ordinary blocks ran 5-10 % slower, and as a
// lambda called from two places the gather build spilled 96 VGPRs (mixed play 2.5x slower).  `continue` ends the block in both.
// No include guard; everything it names is a local of the kernel.
        constexpr bool cp = OC_BLOCK_COOP != 0;
        const long long brow = row0 + static_cast<long long>(blk) * kRowsPerBlock;
        const bool traced = kPolicyTrace && a.trace != nullptr && !trace_done;       // diagnostics build: this wavefront's first block of the launch
        const bool traced2 = kPolicyTrace && a.trace != nullptr && trace_done && !trace2_done;   // ... and, finer, its second
        if (trace_done) trace2_done = true;
        trace_done = true;
        if (traced2) stamp(a, wave, lane, 16);
        if (traced) stamp(a, wave, lane, 2);
        if (cp && kCoopFlags && !in_coop) continue;                   // (eight-wavefront builds: the other wavefronts take no part, no barrier to pass)
        if (!cp || leader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this block's rows have landed (LDS-DMA; a cooperative block: the leader's request)
        if (cp) {
            if (kCoopFlags) {                                         // hand-shake words of group grp: +0 rows in (stretch number), +4 partial sums written (count), +8 sums read (stretch number)
                if (leader) lds_flag_store(lds, a.lds_flag_offset + 16 * grp, coop_seq);
                else lds_flag_wait(lds, a.lds_flag_offset + 16 * grp, coop_seq);
            } else {
                asm volatile("s_barrier" ::: "memory");               // the leaders' rows are in (a bare barrier: see the network barrier above)
                if (!in_coop) { __syncthreads(); __syncthreads(); continue; }   // (the two barriers of the hand-over below)
            }
            rowp = lds + a.lds_weight_bytes + static_cast<size_t>(grp * gsz) * a.lds_wave_bytes + col * RS + 16 * h;   // the leader's staging area
        }
        wave_lds_sync();
        if (FOLD && fold_r1 >= 0) { fold_arrive(fold_r0, fold_r1); fold_r1 = -1; }   // (the previous actor block's actions have completed with that wait)
        const int myrow = myrow_next;
        if (gather && (RB & 15)) {
            // the one row whose last piece could not over-read (the tensor's last row, see stage_rows_gather): its final bytes again,
            // dword by dword -- at most one row per launch, so this branch is all but never taken
            const bool fix = h == 0 && col + blk * kRowsPerBlock < nrows && myrow == static_cast<int>(S.n_rows - 1);
            if (__ballot(fix)) {
                if (fix) {
                    const int tail0 = (RB >> 4) << 4;
                    const uint32_t *sp = reinterpret_cast<const uint32_t *>(S.obs + static_cast<long long>(myrow) * RB + tail0);
                    uint32_t *dp = reinterpret_cast<uint32_t *>(ldsw + col * RS + tail0);
                    for (int q = 0; q < (RB & 15) >> 2; ++q) dp[q] = sp[q];
                }
                wave_lds_sync();
            }
        }
        if (traced) stamp(a, wave, lane, 3);
        if (traced2) stamp(a, wave, lane, 17);
        // A tile with a select code skips the blocks none of whose rows it would write (mixed play: the stretch of
        // kitchens already forced onto the other policy costs this network nothing).
        if (((flags >> 8) & 0xff) && S.row_select) {
            const int want = ((flags >> 8) & 0xff) - 1;
            const bool mine = col + blk * kRowsPerBlock < nrows && S.row_select[brow + col] == want;
            if (__ballot(mine) == 0) {
                if (blk + nwaves < last) stage(blk + nwaves);         // (tiles with a select code have no cooperative block)
                continue;
            }
        }

        // The position loop is software-pipelined by hand: one wavefront per SIMD issues in order, so anything
        // that is not placed in the shadow of an MFMA costs its full latency.  While the 32 fc1 MFMAs of position
        // p run, the VALU turns the conv accumulators of position p+1 into fc1 operands (needed one iteration
        // later, so nothing waits for them), and the LDS reads for position p+2 are in flight.
        // (a cooperative block: this wavefront's positions are [p_first, p_end), the leader's run shorter by about the dense tail it
        // runs alone afterwards -- coop_lead positions, see the kernel; otherwise all of them)
        const int coop_rest = P - coop_lead, coop_each = cp ? (gsz == 2 ? coop_rest : (coop_rest * 21846) >> 16) : 0,     /* coop_rest / (gsz - 1), gsz - 1 = 1 or 3 (exact below 16 384) */ coop_more = cp ? coop_rest - coop_each * (gsz - 1) : 0;
        const int p_first = !cp ? 0 : member == 0 ? 0 : coop_lead + (member - 1) * coop_each + min(member - 1, coop_more);
        const int p_end = !cp ? P : member == 0 ? coop_lead : p_first + coop_each + (member - 1 < coop_more ? 1 : 0);
        v16f h1[2];                                                   // accumulators start at the bias (a cooperative block: the leader's do)
        {
            const v4f *bq = reinterpret_cast<const v4f *>(b1);
            const bool with_bias = !cp || member == 0;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[m][r] = with_bias ? bq[m * 8 + (r >> 2)][r & 3] : 0.f;
        }
        // wave tiles: the dense layers' biases come from L2 -- requested here, used after the position loop
        v4f b2q[TAIL_LDS ? 1 : 8], bhq = {0.f, 0.f, 0.f, 0.f};
        if (!TAIL_LDS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) b2q[TAIL_LDS ? 0 : i] = reinterpret_cast<const v4f *>(b2)[(i >> 2) * 8 + (i & 3)];
            bhq = *reinterpret_cast<const v4f *>(bh);
        }
        // fc1 fragments of a position [piece][k-group][out half].  They are requested one position ahead, into the registers
        // the MFMAs of the current position have just read, so an L2 round trip (fc1 not LDS-resident) has the operand
        // conversion and the next conv to land in.
        v4i w1r[kPieces][2][2];
        auto load_w1 = [&](int p) {
#pragma unroll
            for (int q = 0; q < kPieces; ++q)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo)
                        w1r[q][g][mo] = FC1_LDS ? w1l[(p * (4 * kPieces) + (q * 2 + g) * 2 + mo) * 64]
                                                : w1_from_l2(w1g, lane16, p, (q * 2 + g) * 2 + mo);
        };
        // One wavefront per SIMD (conv digits in registers): the position loop pipelined inside the wavefront -- conv(p+1) and fc1(p-1)
        // beside the conversion of position p, on two sets of conv accumulators (policy_mono_loop.h); same instructions on the same
        // values in the same accumulation order as the loop below: bit-identical outputs.  (Cooperative blocks keep the loop below:
        // their members run a few positions each.)
        bool use_mono = false;
        if constexpr (kMono && !cp) {
            use_mono = a.mono != 0;
            if (use_mono) {
#include "policy_mono_loop.h"
            }
        }
        if (!use_mono) {
            int pi = 0, xq = 0, yq = 0;                                   // position whose B operands `bfr` holds
            auto advance = [&]() { if (pi + 1 < P) { ++pi; if (++yq == PH) { yq = 0; ++xq; } } };   // (clamped: re-reads the last)
            for (int i = 0; i < p_first; ++i) advance();
            v4i bfr[6];
            read_bfrags(rowp, __builtin_amdgcn_readfirstlane((xq * H + yq) * OC_POLICY_C), colstep, bfr);
            v16i acc[NDIG];
            conv(acc, bfr);                                               // the first position
            advance();
            read_bfrags(rowp, __builtin_amdgcn_readfirstlane((xq * H + yq) * OC_POLICY_C), colstep, bfr);
            if (traced2) stamp(a, wave, lane, 18);                         // conv of position 0 issued, next fragments requested
            // fc1 operand of a position: pieces [piece][k-group] of its 16 conv outputs, built pair by pair
            auto make_operand = [&](v4i (&dst)[kPieces][2], int j) {      // pair j = accumulator registers 2j, 2j + 1
                int hi, lo;
                float x0, x1;
                combine_pair(acc, 2 * j, x0, x1);
                split_pair(x0, x1, hi, lo);
                dst[0][j >> 2][j & 3] = hi; dst[1][j >> 2][j & 3] = lo;
            };
            v4i xb[kPieces][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) make_operand(xb, j);
            if (traced2) stamp(a, wave, lane, 19);

            load_w1(p_first);
            // fc1 partial of position p: h1^T[n][row] += W1[n][o*P + p] * X^T[o][row]: two k-groups x three piece products x two
            // output halves.  Unless it is the last, the next position's operand is built around them: the second k-group's
            // registers (pairs 4..7) into spare registers next to the first group's MFMAs, the first group's (pairs 0..3) in place
            // once its MFMAs have issued.
            auto position = [&](int p, auto last_tag) {
                constexpr bool kLast = decltype(last_tag)::value;
                __builtin_amdgcn_sched_barrier(0);
                if (!kLast) conv(acc, bfr);                               // conv of position p+1
                __builtin_amdgcn_sched_barrier(0);
                if (!kLast) {
                    advance();
                    read_bfrags(rowp, __builtin_amdgcn_readfirstlane((xq * H + yq) * OC_POLICY_C), colstep, bfr);    // position p+2
                }
                __builtin_amdgcn_sched_barrier(0);
                v4i nb1[kPieces];
#pragma unroll
                for (int t = 0; t < kTerms; ++t) {
                    h1[0] = mfma_f16(w1r[kTermW[t]][0][0], xb[kTermX[t]][0], h1[0]);
                    h1[1] = mfma_f16(w1r[kTermW[t]][0][1], xb[kTermX[t]][0], h1[1]);
                    if (!kLast) {                                          // pairs 4..7 of the next operand, spread over the three terms: 2, 1, 1
#pragma unroll
                        for (int u = (t == 0 ? 0 : t + 1); u < (t == 0 ? 2 : t + 2); ++u) {
                            int hi, lo;
                            float x0, x1;
                            combine_pair(acc, 8 + 2 * u, x0, x1);
                            split_pair(x0, x1, hi, lo);
                            nb1[0][u] = hi; nb1[1][u] = lo;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < kTerms; ++t) {
                    h1[0] = mfma_f16(w1r[kTermW[t]][1][0], xb[kTermX[t]][1], h1[0]);
                    h1[1] = mfma_f16(w1r[kTermW[t]][1][1], xb[kTermX[t]][1], h1[1]);
                }
                if (!kLast) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_w1(p + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) make_operand(xb, t);
#pragma unroll
                    for (int q = 0; q < kPieces; ++q) xb[q][1] = nb1[q];
                }
                if (traced2 && p < 6) stamp(a, wave, lane, 20 + p);
            };
            {
                int p = p_first;
                for (; p + 1 < p_end; ++p) position(p, std::false_type());
                position(p, std::true_type());
            }
        }
        if (cp) {
            // the hand-over: partial sums of the other members -> the leader, through the members' own staging areas (idle: the stretch
            // is over for them), [register quad][lane] 16 bytes each
            const int fo = a.lds_flag_offset + 16 * grp;
            if (member != 0) {
                v4f *pp = reinterpret_cast<v4f *>(ldsw) + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) pp[i * 64] = v4f{h1[i >> 2][4 * (i & 3)], h1[i >> 2][4 * (i & 3) + 1], h1[i >> 2][4 * (i & 3) + 2], h1[i >> 2][4 * (i & 3) + 3]};
            }
            if (kCoopFlags) {
                if (member != 0) {
                    lds_flag_add(lds, fo + 4, 1, lane);                    // (the release fence in front of the add waits for the stores above)
                    lds_flag_wait(lds, fo + 8, coop_seq);             // the leader has read them: the staging area is this wavefront's again
                    continue;
                }
                lds_flag_wait(lds, fo + 4, gsz - 1);
            } else {
                __syncthreads();
            }
            if (member == 0)
                for (int m = 1; m < gsz; ++m) {
                    const v4f *qq = reinterpret_cast<const v4f *>(lds + a.lds_weight_bytes + static_cast<size_t>(grp * gsz + m) * a.lds_wave_bytes) + lane;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const v4f t = qq[i * 64];
#pragma unroll
                        for (int j = 0; j < 4; ++j) h1[i >> 2][4 * (i & 3) + j] += t[j];
                    }
                }
            if (kCoopFlags) {
                __hip_atomic_store(reinterpret_cast<int *>(lds + fo + 4), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (ready for the next stretch; LDS requests are served in order)
                lds_flag_store(lds, fo + 8, coop_seq);                // (the release fence waits for the reads above)
            } else {
                __syncthreads();                                      // (the members' areas are theirs again)
                if (member != 0) continue;
            }
        }
        if (traced) stamp(a, wave, lane, 4);
        if (traced2) stamp(a, wave, lane, 26);
        // (diagnostics build, timing only -- outputs are then missing: OC_POLICY_DESYNC = -3 ends the block here, -2 behind fc2, -1 behind
        // the head: what the dense tail and the epilogue cost the launch, tools/exp_policy_stagger.py)
        if (kPolicyTrace && a.desync == -3) {
            wave_lds_sync();
            const int nxt3 = cp ? -1 : blk + nwaves < last_full ? blk + nwaves : leader ? last_full + grp : -1;
            if (nxt3 >= 0) stage(nxt3);
            if (h1[0][0] == 12345.678f) S.logp[0] = h1[1][3];          // (keeps the position loop's results alive)
            continue;
        }
        v16f h2[2], out, out2;
        {
            const v4f *bq = reinterpret_cast<const v4f *>(b2);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h2[m][r] = TAIL_LDS ? bq[m * 8 + (r >> 2)][r & 3] : b2q[TAIL_LDS ? 0 : m * 4 + (r >> 2)][r & 3];
            const v4f bhv = TAIL_LDS ? *reinterpret_cast<const v4f *>(bh) : bhq;
#pragma unroll
            for (int r = 0; r < 16; ++r) { out[r] = r < 4 ? bhv[r] : 0.f; out2[r] = 0.f; }
        }
        // the staging area is free: start the copy of this wavefront's next block under the dense layers
        // (all at once: spreading the LDS-DMA instructions between the MFMA groups of fc2 / head was measured -- each still holds
        // the wavefront for its ~0.1 us, the time just moves there: 99.4 vs 95.2 us per cramped_room launch)
        wave_lds_sync();
        fold_drain();                                                 // (the staging area is free: env steps of groups this wavefront completed)
        {
            const int nxt = cp ? -1 : blk + nwaves < last_full ? blk + nwaves : leader ? last_full + grp : -1;   // (a leader's last prefetch: the shared block)
            if (nxt >= 0) stage(nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (traced2) stamp(a, wave, lane, 27);

        // ---- fc2: h2^T[n2][row] = b2 + W2 relu(h1): four k-groups (input half kg >> 1, registers 8 (kg & 1) ..) ----
        // the operand pieces of k-group kg + 1 are built under the MFMAs of k-group kg
        auto split_group = [&](const v16f &v, int g, float descale, v4i (&dst)[kPieces]) {  // descale (a power of two: the layer's weight up-scale off again), relu + clamp, split of registers 8g .. 8g + 7
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int hi, lo;
                const float t0 = v[8 * g + 2 * j] * descale, t1 = v[8 * g + 2 * j + 1] * descale;   // (plain multiplies: see combine_pair)
                split_pair(relu_clamp(t0), relu_clamp(t1), hi, lo);
                dst[0][j] = hi; dst[1][j] = lo;
            }
        };
        // weight fragments: from LDS one k-group ahead; from L2 (wave tiles) all four k-groups at once, and the head's as well,
        // half-way through fc2 -- twelve MFMAs (0.2 us) do not cover an L2 round trip
        constexpr int WB = TAIL_LDS ? 2 : 4;                          // fragment buffers
        v4i whr[WB][kPieces];
        auto load_wh = [&](int kg) {
#pragma unroll
            for (int q = 0; q < kPieces; ++q) whr[kg % WB][q] = head_lane ? wh[(q * 4 + kg) * 16] : v4i{0, 0, 0, 0};
        };
        {
            v4i hb[2][kPieces];                                       // [parity of kg][piece]
            v4i w2r[WB][kPieces][2];                                  // [kg % WB][piece][out half]
            auto load_w2 = [&](int kg) {
#pragma unroll
                for (int q = 0; q < kPieces; ++q)
#pragma unroll
                    for (int mo = 0; mo < 2; ++mo) w2r[kg % WB][q][mo] = w2[((q * 4 + kg) * 2 + mo) * 64];
            };
            load_w2(0);
            if (!TAIL_LDS) {
#pragma unroll
                for (int kg = 1; kg < 4; ++kg) load_w2(kg);
            }
            split_group(h1[0], 0, ds1, hb[0]);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                if (TAIL_LDS && kg + 1 < 4) load_w2(kg + 1);
                if (!TAIL_LDS && kg == 2) {                           // (the first two k-groups' fragments have been consumed)
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) load_wh(k2);
                }
#pragma unroll
                for (int t = 0; t < kTerms; ++t) {
                    h2[0] = mfma_f16(w2r[kg % WB][kTermW[t]][0], hb[kg & 1][kTermX[t]], h2[0]);
                    h2[1] = mfma_f16(w2r[kg % WB][kTermW[t]][1], hb[kg & 1][kTermX[t]], h2[1]);
                }
                if (kg + 1 < 4) split_group(h1[(kg + 1) >> 1], (kg + 1) & 1, ds1, hb[(kg + 1) & 1]);
            }
        }
        if (traced) stamp(a, wave, lane, 5);
        if (traced2) stamp(a, wave, lane, 28);
        if (kPolicyTrace && a.desync == -2) { if (h2[0][0] == 12345.678f) S.logp[0] = h2[1][3]; continue; }
        // ---- head: out^T[a][row], a = reg + 4h for reg 0..3; two accumulators so consecutive MFMAs do not wait for each other ----
        {
            v4i gb[2][kPieces];
            if (TAIL_LDS) load_wh(0);
            split_group(h2[0], 0, ds2, gb[0]);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                if (TAIL_LDS && kg + 1 < 4) load_wh(kg + 1);
                // (twelve MFMAs in all: alternating between two accumulators so consecutive ones do not wait for each other)
#pragma unroll
                for (int t = 0; t < kTerms; ++t) {
                    if ((kg * kTerms + t) & 1) out2 = mfma_f16(whr[kg % WB][kTermW[t]], gb[kg & 1][kTermX[t]], out2);
                    else out = mfma_f16(whr[kg % WB][kTermW[t]], gb[kg & 1][kTermX[t]], out);
                }
                if (kg + 1 < 4) split_group(h2[(kg + 1) >> 1], (kg + 1) & 1, ds2, gb[(kg + 1) & 1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = (out[r] + out2[r]) * ds3;         // the head's weight and operand scales off again
        }
        if (traced) stamp(a, wave, lane, 6);
        if (traced2) stamp(a, wave, lane, 29);
        if (kPolicyTrace && a.desync == -1) { if (out[0] == 12345.678f) S.logp[0] = out[1]; continue; }

        // ---- epilogue.  Lane (col, h) holds outputs a = 4h + r (r = 0..3) of row `col`: the two halves of a row
        //      work on their own actions and meet through three lane exchanges ----
        const long long row = gather ? myrow : brow + col;
        bool valid = (blk * kRowsPerBlock + col) < nrows;
        const int sel = (flags >> 8) & 0xff;                          // OC_POLICY_TILE_SELECT(v): only rows with row_select == v
        if (sel && S.row_select && valid) valid = S.row_select[row] == sel - 1;
        if (!(flags & OC_POLICY_TILE_ACT)) {
            if (valid && S.raw) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * h + r < n_out) S.raw[row * n_out + 4 * h + r] = out[r];
            }
            continue;
        }
        // actor: log-softmax (Categorical(logits=).logits, reference distributions.py:60-80), then argmax (mode) or a draw
        // (Categorical.sample(), distributions.py:20-25) by inverse CDF: ONE uniform per row from the counter hash, the action is the
        // first whose cumulative mass exceeds u x total.  (Rounds 1-3 drew by Gumbel-max: six hashes and twelve logarithms per row --
        // quarter-rate multiplies and transcendentals on a vector pipe that is this kernel's binding resource; the launch with
        // sampled actions took 1.3-1.4 us longer than with argmax)
        // (branch-free over the lane's four outputs: the ones past n_out count as -inf -- exp gives 0, the key never wins -- so
        // the lane halves, which own different numbers of valid outputs, run one straight instruction stream)
        float x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = 4 * h + r < n_out ? out[r] : -INFINITY;
        float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        mx = fmaxf(mx, xor32(mx, h != 0));
        float ex[4], own = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ex[r] = __expf(x[r] - mx); own += ex[r]; }   // v_exp_f32: relative error ~1e-7 on arguments <= 0
        const float other = xor32(own, h != 0);
        const float sum = own + other;                                // (the same value in both halves: the addition commutes)
        const float lse = mx + __logf(sum);                          // sum in [1, n_out]: absolute error ~2e-7
        float lps[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) lps[r] = x[r] - lse;
        if (valid && S.logp) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * h + r < n_out) S.logp[row * n_out + 4 * h + r] = lps[r];
        }
        int best = 0;
        float best_lp = 0.f;
        if (a.sample) {
            // cumulative masses of this half's actions (half 1 starts behind half 0's; an action past n_out adds nothing).  The
            // threshold is u x the LAST cumulative value of that very chain -- half 1's, which `sum` = own + other equals only up to
            // rounding -- and u <= 1 - 2^-24 keeps the f32 product strictly below it: the count therefore always stops at an action
            // whose own mass moved the chain past the threshold (mass > 0), never at one behind the last such (the clamp below is
            // unreachable; with u x sum a threshold rounded up to the chain's end selected action n_out - 1 whatever its mass).
            const float u = oc_policy_uniform_from_hash(oc_policy_hash(S.seed, counter, static_cast<uint32_t>(row), 0));
            float c = h ? other : 0.f, cum[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { c += ex[r]; cum[r] = c; }
            const float c_other = xor32(c, h != 0);                    // (every lane takes part in the exchange)
            const float thr = u * (h ? c : c_other);
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) cnt += cum[r] <= thr ? 1 : 0;
            best = min(cnt + xor32(cnt, h != 0), n_out - 1);
            const int local = best - 4 * h;
            float lp_sel = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) lp_sel = local == r ? lps[r] : lp_sel;
            best_lp = lp_sel + xor32(lp_sel, h != 0);               // (the other half contributes 0)
        } else {
            float best_key = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool wins = lps[r] > best_key;
                best_key = wins ? lps[r] : best_key; best = wins ? 4 * h + r : best; best_lp = wins ? lps[r] : best_lp;
            }
            const float okey = xor32(best_key, h != 0), olp = xor32(best_lp, h != 0);
            const int obest = xor32(best, h != 0);
            if (okey > best_key || (okey == best_key && obest < best)) { best = obest; best_lp = olp; }
        }
        if (valid && h == 0) {
            // FOLD: another XCD's wavefront may read the action within this launch: an agent-scope store (written through its L2)
            if (FOLD && S.actions) __hip_atomic_store(S.actions + row, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (S.actions) S.actions[row] = best;
            if (S.action_logp) S.action_logp[row] = best_lp;
        }
        // FOLD: this block's arrival is announced at the next landing wait (or at the end), when its action stores have completed anyway
        if (FOLD) { fold_r0 = brow; fold_r1 = brow + min(kRowsPerBlock, nrows - blk * kRowsPerBlock) - 1; }
        if (traced) stamp(a, wave, lane, 7);
        if (traced2) stamp(a, wave, lane, 30);
