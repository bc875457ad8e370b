// policy_pack.h -- packed weight image of one policy network for oc_policy_forward.
//
// The network is the reference's CNNBase + one Linear head (train/MAPPO/utils/cnn.py:12-60,
// r_actor_critic.py:11-76 actor / :139-199 critic; hidden_size 64 as train/train_xp.sh:12 sets it):
//   conv 3x3 valid, C=20 -> O=32 channels over the (W, H) grid, ReLU, flatten (o, x', y'), Linear -> 64, ReLU,
//   Linear 64 -> 64, ReLU, Linear 64 -> n_out (6 action logits, or 1 value).
// The kernel evaluates it TRANSPOSED on the matrix cores (activations: feature index on the accumulator
// registers, kitchen row on the lanes), so every weight matrix is the A operand of an MFMA and is stored here in
// exactly the per-lane order the instruction wants; the kernel then reads "fragment f, lane l" with one
// coalesced load and never shuffles weights.  These element functions are shared by the host packer
// (oc_policy_pack_host, CPU tests) and the device packer (oc_policy_pack, stream-ordered).
//
// Lane maps used (cdna_hip_programming.md section 3):
//   C/D of every 32x32 MFMA:  column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), reg 0..15
//   A/B of v_mfma_f32_32x32x16_f16: lane l holds 8 f16 (16 bytes): A[row l & 31][k = 8 * (l >> 5) + i], B likewise by column
//   A/B of v_mfma_i32_32x32x32_i8: lane l holds 16 bytes
//   for both, the pairing of A element i / lane half h with B element i / lane half h is all the packing relies on
//   (which k a slot stands for is free, as long as A and B agree).
//
// Dense layers (fc1, fc2, head): f32 x f32 products on the f16 matrix pipe (round 4; rounds 2-3 used three bf16 pieces and six
// products).  Every f32 value v is split into TWO f16 pieces by round-to-nearest, hi = f16(v), lo = f16(v - hi): 11 + 11
// significant bits plus the sign of lo, i.e. v = hi + lo to 2^-24 of |v| -- f32's own precision -- as long as lo stays a NORMAL
// f16 number.  f16 has five exponent bits, so that is arranged with powers of two, which cost nothing in accuracy:
//   * activations are fed at 2^OC_POLICY_ACT_LOG2 (= 16) times their value and clamped to f16's largest finite number (65504, i.e.
//     activations beyond 4094 saturate; relu and the clamp are one v_med3): full precision for values >= 1/64, an absolute error of
//     2^-29 below;
//   * every layer's weights are scaled up by a power of two of their own (oc_policy_scales: the largest weight of the layer lands in
//     [2^13, 2^14)), the biases follow, and the kernel takes the factor off again with one packed multiply where it turns a layer's
//     result into the next layer's operand (the descale factors are the image's last section).
// w * x is then the THREE piece products hi*hi + hi*lo + lo*hi accumulated in f32 (lo*lo, below 2^-24 of |w x|, is dropped): 3 x 32
// cycles per 32x32x16 tile where the bf16 form took 6 x 32 and v_mfma_f32_32x32x2_f32 8 x 64, and 4 VALU instructions per activation
// pair for the split (two packed conversions, two v_fma_mix for the exact remainders) instead of 7.  Measured against float64
// arithmetic on the same weights the outputs are as close as the reference's own f32 forward (tools/probes/f16_split_accuracy.py).
// An activation register r of lane half h (D layout of the previous layer) goes to B slot i = r & 7 of k-group r >> 3;
// the weight fragments below are stored to match, so a layer's result feeds the next with no lane movement.
#ifndef OC_POLICY_PACK_H
#define OC_POLICY_PACK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define OC_HD __host__ __device__
#else
#define OC_HD
#endif

#define OC_POLICY_MAGIC 0x33504F43          /* "COP3" */
#define OC_POLICY_C 20
#define OC_POLICY_O 32                      /* conv channels = hidden / 2 */
#define OC_POLICY_HID 64
#define OC_POLICY_SLAB 64                   /* k per kernel column kx: 3 tiles x 20 channels = 60, padded */
#define OC_POLICY_HEADER_BYTES 64
#define OC_POLICY_PIECES 2                  /* f16 pieces per f32 value in the dense layers */
#define OC_POLICY_FC1_POS_BYTES (OC_POLICY_PIECES * 2 * 2 * 64 * 16)   /* fc1 fragments of one conv position */
#define OC_POLICY_ACT_LOG2 4                /* dense-layer operands are 2^4 x the activations (f16 range, see above) */
#define OC_POLICY_F16_MAX 65504.0f
/* conv weights are fixed-point numbers of `digits` signed int8 digits in this base: digit 0 (the top one) in [-127, 127], the others
 * balanced in [-BASE/2, BASE/2 - 1].  256 uses every bit of an int8 MFMA operand: 3 digits = 24 bits, the mantissa of the f32 weights
 * themselves (round 2 used base 128: 4 digits = 28 bits at 24 integer MFMAs per conv position instead of 18). */
#define OC_POLICY_DIGIT_BASE 256

// Byte offsets of the sections of one packed network (all multiples of 16).
struct oc_policy_layout {
    int32_t W, H, P, PH;                    // P = (W-2)*(H-2) conv positions, PH = H-2
    int32_t digits, n_out;
    int64_t off_conv, off_conv_sb, off_fc1, off_b1, off_fc2, off_b2, off_head, off_bh, off_ds, total;
};

OC_HD inline oc_policy_layout oc_policy_make_layout(int W, int H, int digits, int n_out) {
    oc_policy_layout L;
    L.W = W; L.H = H; L.PH = H - 2; L.P = (W - 2) * (H - 2); L.digits = digits; L.n_out = n_out;
    int64_t o = OC_POLICY_HEADER_BYTES;
    L.off_conv = o;    o += (int64_t)6 * digits * 64 * 16;         // [kx*2+s][digit][lane][16] int8
    L.off_conv_sb = o; o += 2 * 2 * 16 * 4;                        // scale[h][reg] f32 (folded into fc1; kept for inspection), bias[h][reg] int32 in units of the scale
    L.off_fc1 = o;     o += (int64_t)L.P * OC_POLICY_FC1_POS_BYTES; // [p][piece][k-group 2][out half 2][lane][8] f16
    L.off_b1 = o;      o += 2 * 2 * 16 * 4;                        // [m][h][reg] f32, in the units of the fc1 accumulators
    L.off_fc2 = o;     o += OC_POLICY_PIECES * 4 * 2 * 64 * 16;    // [piece][k-group 4][out half 2][lane][8] f16
    L.off_b2 = o;      o += 2 * 2 * 16 * 4;
    L.off_head = o;    o += OC_POLICY_PIECES * 4 * 16 * 16;        // [piece][k-group 4][slot = 8h + a][8] f16, outputs a < 8 only
    L.off_bh = o;      o += 2 * 4 * 4;                             // [h][reg 0..3] f32
    L.off_ds = o;      o += 8 * 4;                                 // descale factors f32 {fc1 -> fc2 operand, fc2 -> head operand, head -> output, 0}, then the three exponents (int32), 0
    L.total = (o + 255) / 256 * 256;
    return L;
}

// Source tensors in the reference's state-dict layout (f32, contiguous).
struct oc_policy_weights {
    const float *conv_w;    // [O][C][3][3]   base.cnn.cnn.0.weight : multiplies tile (x'+kx, y'+ky), channel c
    const float *conv_b;    // [O]
    const float *fc1_w;     // [64][O*P]      base.cnn.cnn.3.weight : column o*P + p
    const float *fc1_b;     // [64]
    const float *fc2_w;     // [64][64]       base.cnn.cnn.5.weight
    const float *fc2_b;     // [64]
    const float *head_w;    // [n_out][64]    act.action_out.linear.weight / v_out.weight
    const float *head_b;    // [n_out]
};

// accumulator register / lane half -> row of a 32x32 MFMA result
OC_HD inline int oc_mfma_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Largest |w| of conv output channel o: the fixed-point scale of that channel.
OC_HD inline float oc_policy_conv_absmax(const float *conv_w, int o) {
    float m = 0.f;
    for (int i = 0; i < OC_POLICY_C * 9; ++i) {
        float v = conv_w[o * OC_POLICY_C * 9 + i];
        v = v < 0.f ? -v : v;
        if (v > m) m = v;                    // NaN never wins: a NaN weight quantises to 0 digits
    }
    return m;
}

OC_HD inline long long oc_policy_qmax(int digits) {
    long long q = 126;
    for (int i = 1; i < digits; ++i) q *= OC_POLICY_DIGIT_BASE;
    return q;
}

// Fixed-point scale of conv output channel o: one unit of the lowest weight digit.  absmax / qmax uses every bit of the
// digits; a bias that large next to the weights that it would not fit the 32-bit accumulator in these units (|b| > ~4 absmax)
// coarsens the unit instead -- the channel's value is then dominated by the bias, and the relative error stays 2^-24 class.
OC_HD inline float oc_policy_conv_scale(const oc_policy_weights &w, int o, int digits) {
    const float m = oc_policy_conv_absmax(w.conv_w, o);
    float b = w.conv_b[o];
    b = b < 0.f ? -b : b;
    if (!(b == b) || b > 3.0e38f) b = 0.f;                              // NaN / inf bias: the channel is garbage either way
    double s = (double)m / (double)oc_policy_qmax(digits);
    const double sb = (double)b / 1073741824.0;                         // |bias| <= 2^30 units
    if (sb > s) s = sb;
    const float f = (float)s;
    return f > 0.f ? f : 1.0f;                                          // all-zero channel: any unit does
}
// bias of channel o in units of its scale: the initial value of the lowest digit's accumulator
OC_HD inline int32_t oc_policy_conv_bias_units(const oc_policy_weights &w, int o, int digits) {
    const float b = w.conv_b[o];
    if (!(b == b)) return 0;
    double x = (double)b / (double)oc_policy_conv_scale(w, o, digits);
    if (x > 1073741824.0) x = 1073741824.0;
    if (x < -1073741824.0) x = -1073741824.0;
    return (int32_t)(x < 0 ? x - 0.5 : x + 0.5);
}

// Fixed-point conv weight: w ~ scale * sum_d digit_d * BASE^(digits-1-d), digit 0 the top digit in [-127, 127], the others
// balanced in [-BASE/2, BASE/2 - 1].  With 3 digits and scale = absmax / qmax the quantisation step is absmax * 2^-23 -- the
// spacing of the f32 weights next to the channel's largest, i.e. the class of the reference's own f32 rounding (its conv
// accumulates 180 products, each rounded at 2^-24 of the running sum); 4 digits: absmax * 2^-31.
OC_HD inline int oc_policy_conv_digit(float w, float scale, int digits, int d) {
    if (!(scale > 0.f) || !(w == w)) return 0;
    const long long qmax = oc_policy_qmax(digits);
    double x = (double)w / (double)scale;
    if (x > (double)qmax) x = (double)qmax;
    if (x < -(double)qmax) x = -(double)qmax;
    long long q = (long long)(x < 0 ? x - 0.5 : x + 0.5);
    if (q > qmax) q = qmax;
    if (q < -qmax) q = -qmax;
    const long long B = OC_POLICY_DIGIT_BASE, half = OC_POLICY_DIGIT_BASE / 2;
    for (int i = digits - 1; i > d; --i) {                          // strip the lower digits
        long long r = ((q + half) % B + B) % B - half;              // balanced residue in [-B/2, B/2 - 1]
        q = (q - r) / B;
    }
    if (d == 0) return (int)q;
    return (int)(((q + half) % B + B) % B - half);
}

// ---- element functions: value of packed element `idx` of each section -------------------------------------

// conv digits, idx over [6][digits][64][16]
OC_HD inline int8_t oc_policy_pack_conv(const oc_policy_layout &L, const oc_policy_weights &w, int64_t idx) {
    const int j = (int)(idx & 15), lane = (int)((idx >> 4) & 63);
    const int fd = (int)(idx >> 10), d = fd % L.digits, ks = fd / L.digits;      // ks = kx*2 + s
    const int o = lane & 31, h = lane >> 5, kx = ks >> 1, s = ks & 1;
    const int kl = 32 * s + 16 * h + j;
    if (kl >= 60) return 0;
    const int ky = kl / OC_POLICY_C, c = kl % OC_POLICY_C;
    const float v = w.conv_w[((o * OC_POLICY_C + c) * 3 + kx) * 3 + ky];
    return (int8_t)oc_policy_conv_digit(v, oc_policy_conv_scale(w, o, L.digits), L.digits, d);
}
// conv scale (f32) / bias in scale units (int32), idx over [2 kinds][2 h][16 reg], as 32-bit words
OC_HD inline uint32_t oc_policy_pack_conv_sb(const oc_policy_layout &L, const oc_policy_weights &w, int64_t idx) {
    const int reg = (int)(idx & 15), h = (int)((idx >> 4) & 1), kind = (int)(idx >> 5);
    const int o = oc_mfma_row(reg, h);
    union { float f; uint32_t u; } cv;
    if (kind) return (uint32_t)oc_policy_conv_bias_units(w, o, L.digits);
    cv.f = oc_policy_conv_scale(w, o, L.digits);
    return cv.u;
}
// ---- f16 pieces of an f32 weight: w ~ piece0 + piece1, each rounded to nearest even (bit-level, so that the host and the device
// packer write identical images)
OC_HD inline uint32_t oc_f32_bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
OC_HD inline float oc_bits_f32(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
OC_HD inline uint16_t oc_f16_rne(float f) {
    const uint32_t u = oc_f32_bits(f), a = u & 0x7FFFFFFFu;
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    if (a >= 0x7F800000u) return (uint16_t)(sign | (a > 0x7F800000u ? 0x7E00u : 0x7C00u));       // NaN / inf keep their class
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                     // >= 65520 rounds to inf
    if (a <= 0x33000000u) return sign;                                                           // <= 2^-25 rounds to zero (the tie to even)
    const int e = (int)(a >> 23) - 127;
    const uint32_t m = (a & 0x7FFFFFu) | 0x800000u;                                              // 24-bit significand
    const int shift = e >= -14 ? 13 : 13 + (-14 - e);                                            // bits dropped (subnormals: unit 2^-24)
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    if (e >= -14) return (uint16_t)(sign | (uint16_t)(((uint32_t)(e + 15) << 10) + (q - 0x400u)));   // (a carry out of the mantissa lands in the exponent)
    return (uint16_t)(sign | (uint16_t)q);
}
OC_HD inline float oc_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    if (e == 31u) return oc_bits_f32(sign | 0x7F800000u | (m << 13));
    if (e != 0u) return oc_bits_f32(sign | ((e + 112u) << 23) | (m << 13));
    const float v = (float)m * 5.9604644775390625e-08f;                                          // subnormal: m x 2^-24, exact
    return sign ? -v : v;
}
OC_HD inline uint16_t oc_policy_f16_piece(float w, int piece) {
    const uint16_t hi = oc_f16_rne(w);
    if (piece == 0) return hi;
    return oc_f16_rne(w - oc_f16_to_f32(hi));                     // exact subtraction: the remainder has at most 13 significant bits
}
// 2^e as an f32 (|e| <= 126)
OC_HD inline float oc_pow2f(int e) { return oc_bits_f32((uint32_t)(e + 127) << 23); }
// power-of-two exponent that brings a layer's largest |weight| into [2^13, 2^14): the low f16 piece of every weight within 2^-11 of
// the largest is then a normal f16 number
OC_HD inline int oc_policy_up_exponent(float maxabs) {
    const uint32_t a = oc_f32_bits(maxabs) & 0x7FFFFFFFu;
    if (a == 0u || a >= 0x7F800000u) return 0;                    // all-zero or non-finite layer: any factor does
    int e = 13 - ((int)(a >> 23) - 127);
    if (e > 60) e = 60;
    if (e < -60) e = -60;
    return e;
}
// the conv sums reach fc1 as 2^-shift x (integer sum in units of the channel's scale) = 2^OC_POLICY_ACT_LOG2 x value / |largest weight
// of the channel| up to the 126 / 128 of the top digit: an int32 sum is at most 2^31, so nothing can overflow f16 with 3 or 4 digits
OC_HD inline int oc_policy_conv_shift(int digits) { return 8 * (digits - 1) + 7 - OC_POLICY_ACT_LOG2; }
struct oc_policy_scales {
    int32_t e1, e2, e3;                     // fc1, fc2, head weights are stored x 2^e
};
// activation register (k-group g, slot i, lane half h) -> feature index inside a 32-wide half
OC_HD inline int oc_policy_kslot_row(int g, int i, int h) { return oc_mfma_row(8 * g + i, h); }

// The dense layers' A operands: one 16-byte fragment per lane and MFMA, 8 f16 = k slots i = 0..7 of the lane's half.
// value of fc1 weight (out n, conv channel o, position p) as the kernel needs it, before the layer's power-of-two up-scale: the conv
// channel's fixed-point unit is folded in (the kernel feeds 2^-shift x relu(integer conv sum + bias units))
OC_HD inline float oc_policy_fc1_value(const oc_policy_layout &L, const oc_policy_weights &w, int n, int o, int p) {
    return w.fc1_w[(int64_t)n * (OC_POLICY_O * L.P) + o * L.P + p] * oc_policy_conv_scale(w, o, L.digits) * oc_pow2f(oc_policy_conv_shift(L.digits));
}
// the three layers' exponents from their largest |weight| (host: oc_policy_compute_scales below; device: a reduction kernel)
OC_HD inline oc_policy_scales oc_policy_scales_from_max(float max_fc1, float max_fc2, float max_head) {
    oc_policy_scales sc;
    sc.e1 = oc_policy_up_exponent(max_fc1); sc.e2 = oc_policy_up_exponent(max_fc2); sc.e3 = oc_policy_up_exponent(max_head);
    return sc;
}
inline oc_policy_scales oc_policy_compute_scales(const oc_policy_layout &L, const oc_policy_weights &w) {
    float m1 = 0.f, m2 = 0.f, m3 = 0.f, cs[OC_POLICY_O];
    auto upd = [](float &m, float v) { v = v < 0.f ? -v : v; if (v > m && v < 3.0e38f) m = v; };      // (NaN / inf never win)
    for (int o = 0; o < OC_POLICY_O; ++o) cs[o] = oc_policy_conv_scale(w, o, L.digits) * oc_pow2f(oc_policy_conv_shift(L.digits));
    for (int n = 0; n < OC_POLICY_HID; ++n)
        for (int o = 0; o < OC_POLICY_O; ++o)
            for (int p = 0; p < L.P; ++p) upd(m1, w.fc1_w[(int64_t)n * (OC_POLICY_O * L.P) + o * L.P + p] * cs[o]);
    for (int i = 0; i < OC_POLICY_HID * OC_POLICY_HID; ++i) upd(m2, w.fc2_w[i]);
    for (int i = 0; i < L.n_out * OC_POLICY_HID; ++i) upd(m3, w.head_w[i]);
    return oc_policy_scales_from_max(m1, m2, m3);
}
// fc1, idx (in f16 elements) over [P][2 piece][2 g][2 mo][64 lane][8 i]: out n = 32 mo + (lane & 31), in channel o = row(g, i, h)
OC_HD inline uint16_t oc_policy_pack_fc1(const oc_policy_layout &L, const oc_policy_weights &w, const oc_policy_scales &sc, int64_t idx) {
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63), mo = (int)((idx >> 9) & 1), g = (int)((idx >> 10) & 1);
    const int piece = (int)((idx >> 11) % OC_POLICY_PIECES), p = (int)((idx >> 11) / OC_POLICY_PIECES);
    const int n = 32 * mo + (lane & 31), o = oc_policy_kslot_row(g, i, lane >> 5);
    return oc_policy_f16_piece(oc_policy_fc1_value(L, w, n, o, p) * oc_pow2f(sc.e1), piece);
}
// hidden-layer biases in the units of their layer's accumulators (x 2^e), idx over [2 m][2 h][16 reg]
OC_HD inline float oc_policy_pack_hbias(const float *b, int e, int64_t idx) {
    const int reg = (int)(idx & 15), h = (int)((idx >> 4) & 1), m = (int)(idx >> 5);
    return b[32 * m + oc_mfma_row(reg, h)] * oc_pow2f(e);
}
// fc2, idx over [2 piece][4 kg][2 mo][64 lane][8 i]: k-group kg = 2 mi + g covers input half mi, registers 8g..8g+7
OC_HD inline uint16_t oc_policy_pack_fc2(const oc_policy_weights &w, const oc_policy_scales &sc, int64_t idx) {
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63), mo = (int)((idx >> 9) & 1), kg = (int)((idx >> 10) & 3);
    const int piece = (int)(idx >> 12);
    const int in = 32 * (kg >> 1) + oc_policy_kslot_row(kg & 1, i, lane >> 5);
    return oc_policy_f16_piece(w.fc2_w[(32 * mo + (lane & 31)) * OC_POLICY_HID + in] * oc_pow2f(sc.e2), piece);
}
// head, idx over [2 piece][4 kg][16 slot = 8 h + a][8 i]: outputs a < 8 (rows 8..31 of the MFMA's A operand are zero
// registers in the kernel, not stored)
OC_HD inline uint16_t oc_policy_pack_head(const oc_policy_layout &L, const oc_policy_weights &w, const oc_policy_scales &sc, int64_t idx) {
    const int i = (int)(idx & 7), a = (int)((idx >> 3) & 7), h = (int)((idx >> 6) & 1), kg = (int)((idx >> 7) & 3);
    const int piece = (int)(idx >> 9);
    const int in = 32 * (kg >> 1) + oc_policy_kslot_row(kg & 1, i, h);
    return a < L.n_out ? oc_policy_f16_piece(w.head_w[a * OC_POLICY_HID + in] * oc_pow2f(sc.e3), piece) : (uint16_t)0;
}
// head bias, idx over [2 h][4 reg]: output a = reg + 4 h
OC_HD inline float oc_policy_pack_bh(const oc_policy_layout &L, const oc_policy_weights &w, const oc_policy_scales &sc, int64_t idx) {
    const int a = (int)(idx & 3) + 4 * (int)(idx >> 2);
    return a < L.n_out ? w.head_b[a] * oc_pow2f(sc.e3 + OC_POLICY_ACT_LOG2) : 0.f;
}
// the descale section, 8 words: what the kernel multiplies a layer's accumulators with where it turns them into the next layer's
// operand (fc1 -> fc2, fc2 -> head) or into the outputs (head), then the three exponents themselves
OC_HD inline uint32_t oc_policy_pack_ds(const oc_policy_scales &sc, int64_t idx) {
    switch ((int)idx) {
    case 0: return oc_f32_bits(oc_pow2f(OC_POLICY_ACT_LOG2 - sc.e1));
    case 1: return oc_f32_bits(oc_pow2f(-sc.e2));
    case 2: return oc_f32_bits(oc_pow2f(-(sc.e3 + OC_POLICY_ACT_LOG2)));
    case 4: return (uint32_t)sc.e1;
    case 5: return (uint32_t)sc.e2;
    case 6: return (uint32_t)sc.e3;
    default: return 0u;
    }
}

// Element e of the whole image in units of 4 bytes (the header and conv digits are assembled from bytes).
OC_HD inline uint32_t oc_policy_pack_word(const oc_policy_layout &L, const oc_policy_weights &w, const oc_policy_scales &sc, int64_t word) {
    const int64_t b = word * 4;
    union { float f; uint32_t u; } cv;
    if (b < OC_POLICY_HEADER_BYTES) {
        const int32_t hdr[8] = {OC_POLICY_MAGIC, L.W, L.H, OC_POLICY_HID, L.n_out, L.digits, L.P, (int32_t)L.total};
        return word < 8 ? (uint32_t)hdr[word] : 0u;
    }
    if (b < L.off_conv_sb) {
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i)
            v |= (uint32_t)(uint8_t)oc_policy_pack_conv(L, w, b - L.off_conv + i) << (8 * i);
        return v;
    }
    if (b < L.off_fc1) return oc_policy_pack_conv_sb(L, w, (b - L.off_conv_sb) / 4);
    else if (b < L.off_b1) { const int64_t e = (b - L.off_fc1) / 2; return oc_policy_pack_fc1(L, w, sc, e) | ((uint32_t)oc_policy_pack_fc1(L, w, sc, e + 1) << 16); }
    else if (b < L.off_fc2) cv.f = oc_policy_pack_hbias(w.fc1_b, sc.e1, (b - L.off_b1) / 4);
    else if (b < L.off_b2) { const int64_t e = (b - L.off_fc2) / 2; return oc_policy_pack_fc2(w, sc, e) | ((uint32_t)oc_policy_pack_fc2(w, sc, e + 1) << 16); }
    else if (b < L.off_head) cv.f = oc_policy_pack_hbias(w.fc2_b, sc.e2 + OC_POLICY_ACT_LOG2, (b - L.off_b2) / 4);
    else if (b < L.off_bh) { const int64_t e = (b - L.off_head) / 2; return oc_policy_pack_head(L, w, sc, e) | ((uint32_t)oc_policy_pack_head(L, w, sc, e + 1) << 16); }
    else if (b < L.off_ds) cv.f = oc_policy_pack_bh(L, w, sc, (b - L.off_bh) / 4);
    else if (b < L.off_ds + 32) return oc_policy_pack_ds(sc, (b - L.off_ds) / 4);
    else return 0u;
    return cv.u;
}

// Uniform in (0, 1) for the action draw: a pure function of (seed, counter, row, a), so a test can re-derive every sampled action
// from the returned log-probabilities.  The kernel draws by inverse CDF with the ONE uniform of a = 0 per row (rounds 1-3: Gumbel-max
// over the uniforms of a = 0..5).
OC_HD inline uint32_t oc_policy_hash(uint32_t seed, uint32_t counter, uint32_t row, uint32_t a) {
    uint32_t x = seed ^ (counter * 0x9E3779B1u) ^ (row * 0x85EBCA77u) ^ (a * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    x += counter * 0x27D4EB2Fu + a;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}
OC_HD inline float oc_policy_uniform_from_hash(uint32_t x) {
    // 23 bits: (x >> 9) + 0.5 <= 8388607.5 is exact in f32 (24 significant bits), so u lies in
    // [2^-24, 1 - 2^-24]: never 0, never 1 (with 24 bits the top value rounded to even, i.e. to 1.0)
    return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

#endif  // OC_POLICY_PACK_H
