// policy_mono_loop.h -- the position loop of ONE 32-row block pipelined inside its wavefront (round 6), included by
// policy_block_body.h for the builds that run one wavefront per SIMD (conv digits in registers: every grid but 5x4).  No include
// guard; everything it names is a local of the kernel or of the block body.
//
// The loop of policy_block_body.h issues, per position, 18 conv MFMAs with no vector work beside them, 12 fc1 MFMAs with half of the
// next operand's conversion, and the other half (64 vector instructions) with no matrix work: with a SIMD partner the hardware mixes
// two such streams, alone (grids whose rows need the LDS of eight staging areas for four) the matrix pipe idles under every vector
// stretch -- 0.86 us per position on the 9x5 grid against 0.49 us of MFMAs (profiles/r06_trace_policy_9x5_*.log).  Here the wavefront
// keeps TWO sets of conv accumulators and operand registers, the even and the odd positions', and step p runs
//
//     MFMAs   conv(p+1) -> set (p+1) & 1,   fc1(p-1) from set (p-1) & 1      ||      vector   convert(p): set p & 1
//
// the 128 micro-steps of the conversion dealt out four or five behind every MFMA (the half-step of policy_duo_body.h with both
// "blocks" the same rows).  fc1 accumulates into the one h1 in position order: the same instructions on the same values as the loop
// it replaces, bit-identical outputs.
            {
            typedef std::integral_constant<int, 0> SetA;
            typedef std::integral_constant<int, 1> SetB;
            typedef std::false_type NotLast;
            typedef std::true_type Last;
            // a set's position counters: the position whose B fragments it reads next, its row inside the grid column and its byte offset in
            // a staged row -- wave-uniform, and pinned to scalar registers (readfirstlane: left to itself the compiler computed them per lane,
            // ~35 vector instructions per step in one MFMA gap)
            int mq = 0, myq = 0, moff = 0;
            v4i bfr1[6];                                              // ONE set of B fragments: the next conv's, read behind the fc1 MFMAs of the step before
            v16i acc2[2][NDIG];
            v4i xb2[2][kPieces][2];
            // fc1 fragments [set][piece][k-group][out half]: a set's buffer is refilled with the fragments of position p + 1 right behind the
            // MFMAs of fc1(p - 1) that read it -- a whole step ahead of their use (one buffer refilled behind fc1(p - 1) for fc1(p) leaves an
            // L2 round trip only the next conv's 18 MFMAs to land in: measured slower than the plain loop)
            v4i w1r2[2][kPieces][2][2];
            auto load_w1_set = [&](auto Sc, int p) {
                constexpr int b = decltype(Sc)::value;
#pragma unroll
                for (int q = 0; q < kPieces; ++q)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int mo = 0; mo < 2; ++mo)
                            w1r2[b][q][g][mo] = FC1_LDS ? w1l[(p * (4 * kPieces) + (q * 2 + g) * 2 + mo) * 64]
                                                        : w1_from_l2(w1g, lane16, p, (q * 2 + g) * 2 + mo);
            };
            const int col_wrap = colstep - (PH - 1) * OC_POLICY_C;        // from the last position of a column to the first of the next
            auto adv1 = [&]() {                                       // (clamped: re-reads the last; branch-free)
                const int inc = mq + 1 < P ? 1 : 0;
                const int y = myq + inc, wrap = y == PH ? 1 : 0;
                mq = __builtin_amdgcn_readfirstlane(mq + inc);
                myq = __builtin_amdgcn_readfirstlane(wrap ? 0 : y);
                moff = __builtin_amdgcn_readfirstlane(moff + (inc ? (wrap ? col_wrap : OC_POLICY_C) : 0));
            };
            auto read_b6 = [&](auto Kc) {                             // one sixth of a position's B fragments (kernel column kx = k >> 1, half k & 1)
                constexpr int k = decltype(Kc)::value;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(rowp + (moff + (k >> 1) * colstep + 32 * (k & 1)));
                v4i v;
                v.x = static_cast<int>(q[0]); v.y = static_cast<int>(q[1]); v.z = static_cast<int>(q[2]); v.w = static_cast<int>(q[3]);
                bfr1[k] = v;
            };
            auto read_b_all = [&]() { oc_static_for<6>([&](auto Kc) { read_b6(Kc); }); };
            // the conversion of a set's conv sums as 128 single-instruction micro-steps (policy_duo_body.h: convert_u)
            float cv_t[8][2], cv_u[8][2];                              // (one set converts at a time: shared)
            v2h cv_h[8];
            auto convert_u = [&](auto Sc, auto Uc) {
                constexpr int b = decltype(Sc)::value, u = decltype(Uc)::value;
                constexpr int kB = OC_POLICY_DIGIT_BASE;
                const float sc = oc_pow2f(-oc_policy_conv_shift(NDIG));
                if constexpr (NDIG == 3) {
                    constexpr int g = u >> 6, k = (u & 63) >> 2, j = 4 * g + (u & 3), r = 2 * j, e = k & 1;
                    if constexpr (k < 2) cv_t[j][e] = __builtin_bit_cast(float, acc2[b][0][r + e] * kB + acc2[b][1][r + e]);
                    else if constexpr (k < 4) cv_t[j][e] = static_cast<float>(__builtin_bit_cast(int, cv_t[j][e]));
                    else if constexpr (k < 6) cv_u[j][e] = static_cast<float>(acc2[b][2][r + e]);
                    else if constexpr (k < 8) cv_u[j][e] = cv_u[j][e] * sc;
                    else if constexpr (k < 10) cv_t[j][e] = __builtin_fmaf(cv_t[j][e], kB * sc, cv_u[j][e]);
                    else if constexpr (k < 12) cv_t[j][e] = relu_clamp(cv_t[j][e]);
                    else if constexpr (k == 12) {
                        cv_h[j] = __builtin_convertvector(v2f{cv_t[j][0], cv_t[j][1]}, v2h);
                        xb2[b][0][j >> 2][j & 3] = __builtin_bit_cast(int, cv_h[j]);
                    } else if constexpr (k == 13) {
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(cv_u[j][0]) : "v"(cv_h[j]), "v"(cv_t[j][0]));
                    } else if constexpr (k == 14) {
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(cv_u[j][1]) : "v"(cv_h[j]), "v"(cv_t[j][1]));
                    } else {
                        xb2[b][1][j >> 2][j & 3] = __builtin_bit_cast(int, __builtin_convertvector(v2f{cv_u[j][0], cv_u[j][1]}, v2h));
                    }
                } else {                                              // other digit counts: whole pairs (combine_pair: the same arithmetic), one per 16 micro-steps
                    if constexpr ((u & 15) == 15) {
                        constexpr int j = u >> 4;
                        int hi, lo;
                        float x0, x1;
                        combine_pair(acc2[b], 2 * j, x0, x1);
                        split_pair(x0, x1, hi, lo);
                        xb2[b][0][j >> 2][j & 3] = hi; xb2[b][1][j >> 2][j & 3] = lo;
                    }
                }
            };
            constexpr int kCvtSteps = 128, kConvM = 6 * NDIG, kFc1M = 4 * kTerms;
            auto conv_i = [&](auto Sc, auto Ic) {                     // conv MFMA i of 6 x NDIG: kernel column step i / NDIG, digit i % NDIG
                constexpr int b = decltype(Sc)::value, i = decltype(Ic)::value, ks = i / NDIG, d = i % NDIG;
                if constexpr (ks == 0) {
                    if constexpr (d == NDIG - 1) acc2[b][d] = cbeta;
                    else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[b][d][r] = 0;
                    }
                }
                acc2[b][d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wA[ks][d], bfr1[ks], acc2[b][d], 0, 0, 0);
            };
            auto fc1_i = [&](auto Sc, auto Ic) {                      // fc1 MFMA i of 4 x kTerms, in the order of the loop below: k-group, term, output half
                constexpr int b = decltype(Sc)::value, i = decltype(Ic)::value, m = i & 1, t = (i >> 1) % kTerms, g = (i >> 1) / kTerms;
                h1[m] = mfma_f16(w1r2[b][kTermW[t]][g][m], xb2[b][kTermX[t]][g], h1[m]);
            };
            // step p: X = set of p + 1 and p - 1, Y = set of p.  Every MFMA with what runs under it is a scheduling region of its own.
            auto step = [&](auto Xc, auto Yc, auto last_tag, int p) {
                constexpr bool kLast = decltype(last_tag)::value;
                constexpr int nM = (kLast ? 0 : kConvM) + kFc1M;
                if constexpr (!kLast) adv1();                         // (the position whose fragments are read during this step: p + 2, the next step's conv)
                oc_static_for<nM>([&](auto Ic) {
                    constexpr int i = decltype(Ic)::value;
                    if constexpr (!kLast && i < kConvM) conv_i(Xc, Ic);
                    else fc1_i(Xc, std::integral_constant<int, i - (kLast ? 0 : kConvM)>());
                    oc_static_for<(kCvtSteps * (i + 1)) / nM - (kCvtSteps * i) / nM>([&](auto Sc) { convert_u(Yc, std::integral_constant<int, (kCvtSteps * i) / nM + decltype(Sc)::value>()); });
                    // the next step's B fragments: one sixth behind every second fc1 MFMA (the registers are free: this step's conv has issued)
                    if constexpr (!kLast && i >= kConvM && ((i - kConvM) & 1) == 0 && (i - kConvM) / 2 < 6) read_b6(std::integral_constant<int, (i - kConvM) / 2>());
                    __builtin_amdgcn_sched_barrier(0);
                });
                load_w1_set(Xc, a.mono == 2 ? 0 : min(p + 1, P - 1));   // X's buffer: the fragments of fc1(p + 1), its next  (OC_POLICY_MONO=2, timing only -- wrong outputs: position 0's again, from the vector L1)
                __builtin_amdgcn_sched_barrier(0);
            };
            auto fc1_all = [&](auto Sc) { oc_static_for<kFc1M>([&](auto Ic) { fc1_i(Sc, Ic); }); };

            read_b_all();                                             // position 0
            load_w1_set(SetA(), 0);
            load_w1_set(SetB(), min(1, P - 1));
            oc_static_for<kConvM>([&](auto Ic) { conv_i(SetA(), Ic); });     // conv(0)
            adv1();
            read_b_all();                                             // position 1 (P = 1: position 0 again, never used)
            __builtin_amdgcn_sched_barrier(0);
            if (traced2) stamp(a, wave, lane, 18);
            if (P == 1) {
                oc_static_for<kCvtSteps>([&](auto Uc) { convert_u(SetA(), Uc); });
                __builtin_amdgcn_sched_barrier(0);
                fc1_all(SetA());
            } else {
                oc_static_for<kConvM>([&](auto Ic) {                  // conv(1) beside convert(0)
                    constexpr int i = decltype(Ic)::value;
                    conv_i(SetB(), Ic);
                    oc_static_for<(kCvtSteps * (i + 1)) / kConvM - (kCvtSteps * i) / kConvM>([&](auto Sc) { convert_u(SetA(), std::integral_constant<int, (kCvtSteps * i) / kConvM + decltype(Sc)::value>()); });
                    __builtin_amdgcn_sched_barrier(0);
                });
                adv1();
                read_b_all();                                         // position 2
                __builtin_amdgcn_sched_barrier(0);
                if (traced2) stamp(a, wave, lane, 19);
                int p = 1;
                for (; p + 2 <= P - 1; p += 2) {
                    step(SetA(), SetB(), NotLast(), p);               // conv(p+1), fc1(p-1)  ||  convert(p)        p odd
                    step(SetB(), SetA(), NotLast(), p + 1);           // conv(p+2), fc1(p)    ||  convert(p+1)
                    if (traced2 && p < 12) stamp(a, wave, lane, 20 + (p >> 1));
                }
                if (p == P - 1) {                                     // P even: the last step converts an odd position
                    step(SetA(), SetB(), Last(), p);
                    fc1_all(SetB());
                } else {                                              // P odd
                    step(SetA(), SetB(), NotLast(), p);
                    step(SetB(), SetA(), Last(), p + 1);
                    fc1_all(SetA());
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            }
