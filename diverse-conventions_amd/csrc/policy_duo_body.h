// policy_duo_body.h -- one PAIR of 32-row blocks (bA, bB) of oc_policy_duo_kernel (policy_duo_kernel.h), interleaved; included once
// inside the kernel's item loop (no include guard; everything it names is a local of the kernel).  Arithmetic and the order of every
// accumulation are those of policy_block_body.h: a block's outputs do not depend on whether it ran alone or as half of a pair.
        {
#ifndef OC_DUO_ABLATE
#define OC_DUO_ABLATE 0      /* timing experiments only (tools/ab_duo_ablate.sh): 1 no operand conversion in the position loop, 2 fc1 fragments loaded once, 4 no B-fragment reads in the loop, 8 no conv MFMAs, 16 no fc1 MFMAs -- outputs are then wrong */
#endif
        typedef std::integral_constant<int, 0> BlkA;
        typedef std::integral_constant<int, 1> BlkB;
        // (diagnostics build: phase stamps of this wavefront's first pair of the launch -- coarse -- and of its second -- fine; the slots of
        // tools/trace_policy.py, a pair standing where the one-block kernel has a block)
        const bool traced = kPolicyTrace && a.trace != nullptr && !trace_done;
        const bool traced2 = kPolicyTrace && a.trace != nullptr && trace_done && !trace2_done;
        if (trace_done) trace2_done = true;
        trace_done = true;
        if (traced2) stamp(a, wave, lane, 16);
        if (traced) stamp(a, wave, lane, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // both blocks' rows have landed (LDS-DMA)
        wave_lds_sync();
        if (traced) stamp(a, wave, lane, 3);
        if (traced2) stamp(a, wave, lane, 17);
        const uint8_t *const rp[2] = {ldsw + col * RB + 16 * h, ldsw2 + col * RB + 16 * h};
        // Byte offset (inside a staged row) of the conv position whose B operands a block reads next.  A build for ONE grid (GW > 0:
        // the kernel's template arguments) knows every offset at compile time -- the position loop is unrolled, the LDS reads take
        // immediate offsets off one base register and nothing scalar is left in the loop; the generic build (GW == 0) walks counters.
        constexpr bool kStaticGrid = GW > 0;
        constexpr int kSPH = kStaticGrid ? GH - 2 : 1, kSP = kStaticGrid ? (GW - 2) * (GH - 2) : 1, kSColstep = GH * OC_POLICY_C;
        int pq[2] = {0, 0}, xq[2] = {0, 0}, yq[2] = {0, 0};           // generic build: position whose B operands bfr[b] holds (wave-uniform: scalar registers)
        v4i bfr[2][6];
        v16i acc[2][NDIG];
        v16f h1[2][2];
        v4i xb[2][kPieces][2];                                        // fc1 operand of a block's current position [piece][k-group]
        v4i w1r[kPieces][2][2];                                       // fc1 fragments of ONE position [piece][k-group][out half], read once for both blocks
        auto advance = [&](auto Bc) {
            constexpr int b = decltype(Bc)::value;
            // (clamped: re-reads the last; branch-free -- a branch here would cut the half-step into two scheduling regions)
            const int inc = pq[b] + 1 < P ? 1 : 0;
            pq[b] += inc;
            const int y = yq[b] + inc, wrap = y == PH ? 1 : 0;
            yq[b] = wrap ? 0 : y;
            xq[b] += wrap;
        };
        auto read_b = [&](auto Bc) {
            constexpr int b = decltype(Bc)::value;
            read_bfrags(rp[b], (xq[b] * H + yq[b]) * OC_POLICY_C, colstep, bfr[b]);
        };
        // static grid: offset of position min(p, P - 1) (clamped: the last is re-read)
        auto static_off = [](int p) constexpr { const int q = p < kSP - 1 ? p : kSP - 1; return ((q / kSPH) * GH + q % kSPH) * OC_POLICY_C; };
        (void)static_off;
        // A block's operand conversion as 128 single-instruction micro-steps (NDIG = 3; pair j = accumulator registers 2j, 2j + 1 goes
        // through 16 instructions: top digits combined x2, converted x2, low digit converted x2, scaled x2, fma x2, relu / clamp x2, f16
        // hi, remainder x2, f16 lo), dealt out behind the other block's MFMAs FOUR OR FIVE to a gap: five independent single-issue
        // instructions are what hides under a 32-cycle MFMA and the sixth costs ~22 cycles (tools/probes/mfma_valu_overlap.hip), and a
        // dependent instruction right behind its producer waits for it -- so four pairs advance together, stage by stage: micro-step
        // u = 64 g + 4 k + i is stage k of pair 4 g + i, and the instructions of one gap belong to four different pairs.
        float cv_t[2][8][2], cv_u[2][8][2];
        v2h cv_h[2][8];
        auto convert_u = [&](auto Bc, auto Uc) {
            constexpr int b = decltype(Bc)::value, u = decltype(Uc)::value;
            constexpr int kB = OC_POLICY_DIGIT_BASE;
            const float sc = oc_pow2f(-oc_policy_conv_shift(NDIG));
            if constexpr (NDIG == 3) {
                constexpr int g = u >> 6, k = (u & 63) >> 2, j = 4 * g + (u & 3), r = 2 * j, e = k & 1;
                if constexpr (k < 2) cv_t[b][j][e] = __builtin_bit_cast(float, acc[b][0][r + e] * kB + acc[b][1][r + e]);            // v_lshl_add_u32 (the integer, for now)
                else if constexpr (k < 4) cv_t[b][j][e] = static_cast<float>(__builtin_bit_cast(int, cv_t[b][j][e]));                 // v_cvt_f32_i32
                else if constexpr (k < 6) cv_u[b][j][e] = static_cast<float>(acc[b][2][r + e]);                                      // v_cvt_f32_i32
                else if constexpr (k < 8) cv_u[b][j][e] = cv_u[b][j][e] * sc;                                                        // v_mul_f32
                else if constexpr (k < 10) cv_t[b][j][e] = __builtin_fmaf(cv_t[b][j][e], kB * sc, cv_u[b][j][e]);                    // v_fmac_f32
                else if constexpr (k < 12) cv_t[b][j][e] = relu_clamp(cv_t[b][j][e]);                                                // v_med3_f32
                else if constexpr (k == 12) {                                                                                        // v_cvt_pk_f16_f32: the hi pieces
                    cv_h[b][j] = __builtin_convertvector(v2f{cv_t[b][j][0], cv_t[b][j][1]}, v2h);
                    xb[b][0][j >> 2][j & 3] = __builtin_bit_cast(int, cv_h[b][j]);
                } else if constexpr (k == 13) {                                                                                      // v_fma_mix_f32: x - hi, exact
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(cv_u[b][j][0]) : "v"(cv_h[b][j]), "v"(cv_t[b][j][0]));
                } else if constexpr (k == 14) {
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(cv_u[b][j][1]) : "v"(cv_h[b][j]), "v"(cv_t[b][j][1]));
                } else {                                                                                                             // v_cvt_pk_f16_f32: the lo pieces
                    xb[b][1][j >> 2][j & 3] = __builtin_bit_cast(int, __builtin_convertvector(v2f{cv_u[b][j][0], cv_u[b][j][1]}, v2h));
                }
            } else {                                                  // other digit counts: whole pairs (combine_pair: the same arithmetic), one per 16 micro-steps
                if constexpr ((u & 15) == 15) {
                    constexpr int j = u >> 4;
                    int hi, lo;
                    float x0, x1;
                    combine_pair(acc[b], 2 * j, x0, x1);
                    split_pair(x0, x1, hi, lo);
                    xb[b][0][j >> 2][j & 3] = hi; xb[b][1][j >> 2][j & 3] = lo;
                }
            }
        };
        constexpr int kCvtSteps = 128;
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
        // the MFMAs of a block, one at a time: conv MFMA i of 6 x NDIG (kernel column step ks = i / NDIG, digit d = i % NDIG: a digit's
        // accumulator starts from zero, the lowest digit's from the bias) ...
        auto conv_i = [&](auto Bc, auto Ic) {
            constexpr int b = decltype(Bc)::value, i = decltype(Ic)::value, ks = i / NDIG, d = i % NDIG;
            if constexpr (ks == 0) {
                if constexpr (d == NDIG - 1) acc[b][d] = cbeta;
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[b][d][r] = 0;
                }
            }
            acc[b][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wA[ks][d], bfr[b][ks], acc[b][d], 0, 0, 0);
        };
        // ... fc1 MFMA i of 4 x kTerms, in the order of policy_block_body.h: k-group g, term t, output half m
        auto fc1_i = [&](auto Bc, auto Ic) {
            constexpr int b = decltype(Bc)::value, i = decltype(Ic)::value, m = i & 1, t = (i >> 1) % kTerms, g = (i >> 1) / kTerms;
            h1[b][m] = mfma_f16(w1r[kTermW[t]][g][m], xb[b][kTermX[t]][g], h1[b][m]);
        };
        // one sixth of a block's B fragments (kernel column kx, half s) of the position its counters point at
        auto read_b_i = [&](auto Bc, auto Kc, auto Pc) {              // Pc: the position (static grid; ignored by the generic build)
            constexpr int b = decltype(Bc)::value, k = decltype(Kc)::value;
            const uint32_t *q;
            if constexpr (kStaticGrid) q = reinterpret_cast<const uint32_t *>(rp[b] + (static_off(decltype(Pc)::value) + (k >> 1) * kSColstep + 32 * (k & 1)));
            else q = reinterpret_cast<const uint32_t *>(rp[b] + (xq[b] * H + yq[b]) * OC_POLICY_C + (k >> 1) * colstep + 32 * (k & 1));
            v4i v;
            v.x = static_cast<int>(q[0]); v.y = static_cast<int>(q[1]); v.z = static_cast<int>(q[2]); v.w = static_cast<int>(q[3]);
            bfr[b][k] = v;
        };
        constexpr int kConvM = 6 * NDIG, kFc1M = 4 * kTerms;
        // A half-step: block X's MFMAs (its conv of the next position unless this is the last, then its fc1 of the current one), block Y's
        // conversion quarters dealt out behind them, X's next B fragments behind its conv.  Every MFMA with what runs under it is a
        // scheduling region of its own (sched_barrier): the order below IS the instruction order.
        auto half_step = [&](auto Xc, auto Yc, auto last_tag, auto Pc) {   // Pc (static grid): the position whose B fragments X reads during this half-step
            constexpr bool kLast = decltype(last_tag)::value;
            constexpr int nM = (kLast ? 0 : kConvM) + kFc1M;
            if constexpr (!kLast && !kStaticGrid) advance(Xc);        // (X's counters: the position whose fragments are read during this half-step)
            oc_static_for<nM>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
                if constexpr (!kLast && i < kConvM) { if constexpr (!(OC_DUO_ABLATE & 8)) conv_i(Xc, Ic); }
                else { if constexpr (!(OC_DUO_ABLATE & 16)) fc1_i(Xc, std::integral_constant<int, i - (kLast ? 0 : kConvM)>()); }
                // quarters [32 i / nM, 32 (i + 1) / nM) of Y's conversion
                if constexpr (!(OC_DUO_ABLATE & 1) || kLast)
                    oc_static_for<(kCvtSteps * (i + 1)) / nM - (kCvtSteps * i) / nM>([&](auto Sc) { convert_u(Yc, std::integral_constant<int, (kCvtSteps * i) / nM + decltype(Sc)::value>()); });
                // X's B fragments of the position after next: one sixth behind every second fc1 MFMA (their registers are free: X's conv has issued)
                if constexpr (!(OC_DUO_ABLATE & 4) && !kLast && i >= kConvM && ((i - kConvM) & 1) == 0 && (i - kConvM) / 2 < 6) read_b_i(Xc, std::integral_constant<int, (i - kConvM) / 2>(), Pc);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        typedef std::false_type NotLast;
        typedef std::true_type Last;

        // ---- prologue: both blocks' first conv, A's first operand under B's conv
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const v4f *bq = reinterpret_cast<const v4f *>(b1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) h1[b][m][r] = bq[m * 8 + (r >> 2)][r & 3];
        }
        typedef std::integral_constant<int, 0> Pos0;
        typedef std::integral_constant<int, 1> Pos1;
        auto read_b_all = [&](auto Bc, auto Pc) {
            if constexpr (kStaticGrid) oc_static_for<6>([&](auto Kc) { read_b_i(Bc, Kc, Pc); });
            else read_b(Bc);
        };
        read_b_all(BlkA(), Pos0());
        read_b_all(BlkB(), Pos0());
        load_w1(0);
        oc_static_for<kConvM>([&](auto Ic) { conv_i(BlkA(), Ic); });  // A.conv(0)
        if constexpr (!kStaticGrid) advance(BlkA());
        read_b_all(BlkA(), Pos1());                                   // A: position 1
        __builtin_amdgcn_sched_barrier(0);
        if (traced2) stamp(a, wave, lane, 18);
        oc_static_for<kConvM>([&](auto Ic) {                          // B.conv(0) beside A.convert(0)
            constexpr int i = decltype(Ic)::value;
            conv_i(BlkB(), Ic);
            oc_static_for<(kCvtSteps * (i + 1)) / kConvM - (kCvtSteps * i) / kConvM>([&](auto Sc) { convert_u(BlkA(), std::integral_constant<int, (kCvtSteps * i) / kConvM + decltype(Sc)::value>()); });
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!kStaticGrid) advance(BlkB());
        read_b_all(BlkB(), Pos1());                                   // B: position 1
        __builtin_amdgcn_sched_barrier(0);
        if (traced2) stamp(a, wave, lane, 19);

        // ---- the position loop: per position two half-steps, each the MFMAs of one block beside the other block's operand conversion
        if constexpr (kStaticGrid) {
            oc_static_for<kSP - 1>([&](auto Pc) {
                constexpr int p = decltype(Pc)::value;
                half_step(BlkA(), BlkB(), NotLast(), std::integral_constant<int, p + 2>());     // A.conv(p+1), A.fc1(p)  ||  B.convert(p)
                half_step(BlkB(), BlkA(), NotLast(), std::integral_constant<int, p + 2>());     // B.conv(p+1), B.fc1(p)  ||  A.convert(p+1)
                if (!(OC_DUO_ABLATE & 2)) load_w1(p + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (traced2 && p < 6) stamp(a, wave, lane, 20 + p);
            });
        } else {
            for (int p = 0; p + 1 < P; ++p) {
                half_step(BlkA(), BlkB(), NotLast(), Pos0());         // A.conv(p+1), A.fc1(p)  ||  B.convert(p)
                half_step(BlkB(), BlkA(), NotLast(), Pos0());         // B.conv(p+1), B.fc1(p)  ||  A.convert(p+1)
                if (!(OC_DUO_ABLATE & 2)) load_w1(p + 1);             // the next position's fc1 fragments: requested behind B's fc1, used behind A's next conv
                __builtin_amdgcn_sched_barrier(0);
                if (traced2 && p < 6) stamp(a, wave, lane, 20 + p);
            }
        }

        // ---- the dense tail, interleaved the same way.  A layer input of a block (four k-groups of eight registers) becomes operand pieces
        // in 32 half-pair steps of four vector instructions: pair u = registers 2u, 2u + 1 of the 32 (input half u >> 3); step 2u: descale,
        // relu + clamp; step 2u + 1: the f16 split.
        v4i hb[2][4][kPieces];                                        // a block's operand of the current dense layer [k-group][piece]
        float spl_t[2][2];
        auto split_q = [&](auto Bc, auto Sc, const v16f (&v)[2], float descale) {
            constexpr int b = decltype(Bc)::value, s = decltype(Sc)::value, u = s >> 1, kg = u >> 2, j = u & 3;
            if constexpr ((s & 1) == 0) {
                spl_t[b][0] = relu_clamp(v[kg >> 1][8 * (kg & 1) + 2 * j] * descale);
                spl_t[b][1] = relu_clamp(v[kg >> 1][8 * (kg & 1) + 2 * j + 1] * descale);
            } else {
                int hi, lo;
                split_pair(spl_t[b][0], spl_t[b][1], hi, lo);
                hb[b][kg][0][j] = hi; hb[b][kg][1][j] = lo;
            }
        };
        // nM MFMAs (mfma(i)) with the 32 steps of a conversion (step(s)) dealt out behind them
        auto deal = [&](auto Nc, auto &&mfma, auto &&step) {
            constexpr int nM = decltype(Nc)::value;
            oc_static_for<nM>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
                mfma(Ic);
                oc_static_for<(32 * (i + 1)) / nM - (32 * i) / nM>([&](auto Sc) { step(std::integral_constant<int, (32 * i) / nM + decltype(Sc)::value>()); });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // last position: A.fc1(P-1) || B.convert(P-1);  B.fc1(P-1) || A's fc2 operand
        half_step(BlkA(), BlkB(), Last(), Pos0());
        if (traced2) stamp(a, wave, lane, 25);
        deal(std::integral_constant<int, kFc1M>(), [&](auto Ic) { fc1_i(BlkB(), Ic); }, [&](auto Sc) { split_q(BlkA(), Sc, h1[0], ds1); });

        if (traced) stamp(a, wave, lane, 4);
        if (traced2) stamp(a, wave, lane, 26);
        // the staging areas are free: the next item's rows travel under the dense tail
        wave_lds_sync();
        if (item + 1 < n_items) stage_item(item + 1);
        __builtin_amdgcn_sched_barrier(0);

        v16f h2[2][2], outv[2], out2[2];
        {
            const v4f *bq = reinterpret_cast<const v4f *>(b2);
            const v4f bhv = *reinterpret_cast<const v4f *>(bh);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) h2[b][m][r] = bq[m * 8 + (r >> 2)][r & 3];
#pragma unroll
                for (int r = 0; r < 16; ++r) { outv[b][r] = r < 4 ? bhv[r] : 0.f; out2[b][r] = 0.f; }
            }
        }
        v4i w2r[4][kPieces][2];                                       // fc2 fragments [k-group][piece][out half]: read once for both blocks
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int q = 0; q < kPieces; ++q)
#pragma unroll
                for (int mo = 0; mo < 2; ++mo) w2r[kg][q][mo] = w2[((q * 4 + kg) * 2 + mo) * 64];
        v4i whr[4][kPieces];                                          // head fragments [k-group][piece]
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
            for (int q = 0; q < kPieces; ++q) whr[kg][q] = head_lane ? wh[(q * 4 + kg) * 16] : v4i{0, 0, 0, 0};
        // the epilogue of one block in the lanes that hold its outputs: lane (col, h) holds outputs a = 4h + r (r = 0..3) of row `col`
        // (policy_block_body.h: the same instructions)
        auto epilogue = [&](v16f &out, const v16f &o2, int blk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = (out[r] + o2[r]) * ds3;       // the head's weight and operand scales off again
            const long long row = row0 + static_cast<long long>(blk) * kRowsPerBlock + col;
            const bool valid = (blk * kRowsPerBlock + col) < nrows;
            if (!(flags & OC_POLICY_TILE_ACT)) {
                if (valid && S.raw) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * h + r < n_out) S.raw[row * n_out + 4 * h + r] = out[r];
                }
                return;
            }
            float x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = 4 * h + r < n_out ? out[r] : -INFINITY;
            float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
            mx = fmaxf(mx, xor32(mx, h != 0));
            float ex[4], own = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ex[r] = __expf(x[r] - mx); own += ex[r]; }
            const float other = xor32(own, h != 0);
            const float sum = own + other;
            const float lse = mx + __logf(sum);
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
                best_lp = lp_sel + xor32(lp_sel, h != 0);
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
                if (S.actions) S.actions[row] = best;
                if (S.action_logp) S.action_logp[row] = best_lp;
            }
        };

        // fc2 MFMA i of 8 x kTerms / head MFMA i of 4 x kTerms of a block, in the order of policy_block_body.h
        auto fc2_i = [&](auto Bc, auto Ic) {
            constexpr int b = decltype(Bc)::value, i = decltype(Ic)::value, m = i & 1, t = (i >> 1) % kTerms, kg = (i >> 1) / kTerms;
            h2[b][m] = mfma_f16(w2r[kg][kTermW[t]][m], hb[b][kg][kTermX[t]], h2[b][m]);
        };
        auto head_i = [&](auto Bc, auto Ic) {
            constexpr int b = decltype(Bc)::value, i = decltype(Ic)::value, t = i % kTerms, kg = i / kTerms;
            if constexpr ((i & 1) != 0) out2[b] = mfma_f16(whr[kg][kTermW[t]], hb[b][kg][kTermX[t]], out2[b]);
            else outv[b] = mfma_f16(whr[kg][kTermW[t]], hb[b][kg][kTermX[t]], outv[b]);
        };
        if (traced2) stamp(a, wave, lane, 27);
        // fc2(A) || B's fc2 operand
        deal(std::integral_constant<int, 8 * kTerms>(), [&](auto Ic) { fc2_i(BlkA(), Ic); }, [&](auto Sc) { split_q(BlkB(), Sc, h1[1], ds1); });
        // fc2(B) || A's head operand (its fc2 operand has been consumed: the same registers)
        deal(std::integral_constant<int, 8 * kTerms>(), [&](auto Ic) { fc2_i(BlkB(), Ic); }, [&](auto Sc) { split_q(BlkA(), Sc, h2[0], ds2); });
        if (traced) stamp(a, wave, lane, 5);
        if (traced2) stamp(a, wave, lane, 28);
        // head(A) || B's head operand
        deal(std::integral_constant<int, 4 * kTerms>(), [&](auto Ic) { head_i(BlkA(), Ic); }, [&](auto Sc) { split_q(BlkB(), Sc, h2[1], ds2); });
        // head(B) || A's epilogue
        oc_static_for<4 * kTerms>([&](auto Ic) { head_i(BlkB(), Ic); });
        if (traced) stamp(a, wave, lane, 6);
        if (traced2) stamp(a, wave, lane, 29);
        epilogue(outv[0], out2[0], bA);
        __builtin_amdgcn_sched_barrier(0);
        epilogue(outv[1], out2[1], bB);
        if (traced) stamp(a, wave, lane, 7);
        if (traced2) stamp(a, wave, lane, 30);
        }
