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
//   A of v_mfma_f32_32x32x2_f32: lane l holds A[row l & 31][k = l >> 5]
//   A/B of v_mfma_i32_32x32x32_i8: lane l holds 16 bytes; the pairing of A byte j / lane half h with B byte j /
//     lane half h is all the packing relies on (integer sums do not care about the order of k).
#ifndef OC_POLICY_PACK_H
#define OC_POLICY_PACK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define OC_HD __host__ __device__
#else
#define OC_HD
#endif

#define OC_POLICY_MAGIC 0x31504F43          /* "COP1" */
#define OC_POLICY_C 20
#define OC_POLICY_O 32                      /* conv channels = hidden / 2 */
#define OC_POLICY_HID 64
#define OC_POLICY_SLAB 64                   /* k per kernel column kx: 3 tiles x 20 channels = 60, padded */
#define OC_POLICY_HEADER_BYTES 64

// Byte offsets of the sections of one packed network (all multiples of 16).
struct oc_policy_layout {
    int32_t W, H, P, PH;                    // P = (W-2)*(H-2) conv positions, PH = H-2
    int32_t digits, n_out;
    int64_t off_conv, off_conv_sb, off_fc1, off_b1, off_fc2, off_b2, off_head, off_bh, total;
};

OC_HD inline oc_policy_layout oc_policy_make_layout(int W, int H, int digits, int n_out) {
    oc_policy_layout L;
    L.W = W; L.H = H; L.PH = H - 2; L.P = (W - 2) * (H - 2); L.digits = digits; L.n_out = n_out;
    int64_t o = OC_POLICY_HEADER_BYTES;
    L.off_conv = o;    o += (int64_t)6 * digits * 64 * 16;         // [kx*2+s][digit][lane][16] int8
    L.off_conv_sb = o; o += 2 * 2 * 16 * 4;                        // scale[h][reg], bias[h][reg] f32
    L.off_fc1 = o;     o += (int64_t)L.P * 8 * 64 * 4 * 4;         // [p][q][lane][4] f32, step f = 4q+j = 2s+m
    L.off_b1 = o;      o += 2 * 2 * 16 * 4;                        // [m][h][reg]
    L.off_fc2 = o;     o += 2 * 8 * 64 * 4 * 4;                    // [m][q][lane][4], f = 2s+m2
    L.off_b2 = o;      o += 2 * 2 * 16 * 4;
    L.off_head = o;    o += 8 * 64 * 4 * 4;                        // [q][lane][4], f = 16m+s
    L.off_bh = o;      o += 2 * 4 * 4;                             // [h][reg 0..3]
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
    for (int i = 1; i < digits; ++i) q *= 128;
    return q;
}

// Fixed-point conv weight: w ~ (absmax / qmax) * sum_d digit_d * 128^(digits-1-d), digit 0 the top digit in
// [-127, 127], the others balanced in [-64, 63].  With 4 digits the quantisation step is absmax * 2^-28: below
// the f32 rounding of the reference's own accumulation.
OC_HD inline int oc_policy_conv_digit(float w, float absmax, int digits, int d) {
    if (!(absmax > 0.f) || !(w == w)) return 0;
    const long long qmax = oc_policy_qmax(digits);
    double x = (double)w / (double)absmax * (double)qmax;
    long long q = (long long)(x < 0 ? x - 0.5 : x + 0.5);
    if (q > qmax) q = qmax;
    if (q < -qmax) q = -qmax;
    for (int i = digits - 1; i > d; --i) {                          // strip the lower digits
        long long r = ((q + 64) % 128 + 128) % 128 - 64;            // balanced residue in [-64, 63]
        q = (q - r) / 128;
    }
    if (d == 0) return (int)q;
    return (int)(((q + 64) % 128 + 128) % 128 - 64);
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
    return (int8_t)oc_policy_conv_digit(v, oc_policy_conv_absmax(w.conv_w, o), L.digits, d);
}
// conv scale / bias, idx over [2 kinds][2 h][16 reg]
OC_HD inline float oc_policy_pack_conv_sb(const oc_policy_layout &L, const oc_policy_weights &w, int64_t idx) {
    const int reg = (int)(idx & 15), h = (int)((idx >> 4) & 1), kind = (int)(idx >> 5);
    const int o = oc_mfma_row(reg, h);
    if (kind) return w.conv_b[o];
    const float m = oc_policy_conv_absmax(w.conv_w, o);
    return (float)((double)m / (double)oc_policy_qmax(L.digits));
}
// The dense layers' A operands are stored so that one lane reads four consecutive MFMA steps with one 16-byte
// load: [.. ][q = f / 4][lane][f % 4], f the position of the MFMA in the kernel's issue order.
// fc1, idx over [P][8 q][64 lane][4]: issue order f = 2 s + m (k-step s = 0..15, output half m)
OC_HD inline float oc_policy_pack_fc1(const oc_policy_layout &L, const oc_policy_weights &w, int64_t idx) {
    const int f = (int)(4 * ((idx >> 8) & 7) + (idx & 3)), lane = (int)((idx >> 2) & 63), p = (int)(idx >> 11);
    const int s = f >> 1, m = f & 1;
    const int n = 32 * m + (lane & 31), o = oc_mfma_row(s, lane >> 5);
    return w.fc1_w[(int64_t)n * (OC_POLICY_O * L.P) + o * L.P + p];
}
// hidden-layer biases, idx over [2 m][2 h][16 reg]
OC_HD inline float oc_policy_pack_hbias(const float *b, int64_t idx) {
    const int reg = (int)(idx & 15), h = (int)((idx >> 4) & 1), m = (int)(idx >> 5);
    return b[32 * m + oc_mfma_row(reg, h)];
}
// fc2, idx over [2 m][8 q][64 lane][4]: input half m, issue order inside it f = 2 s + m2
OC_HD inline float oc_policy_pack_fc2(const oc_policy_weights &w, int64_t idx) {
    const int f = (int)(4 * ((idx >> 8) & 7) + (idx & 3)), lane = (int)((idx >> 2) & 63), m = (int)(idx >> 11);
    const int s = f >> 1, m2 = f & 1;
    return w.fc2_w[(32 * m2 + (lane & 31)) * OC_POLICY_HID + 32 * m + oc_mfma_row(s, lane >> 5)];
}
// head, idx over [8 q][64 lane][4]: issue order f = 16 m + s; rows >= n_out are zero
OC_HD inline float oc_policy_pack_head(const oc_policy_layout &L, const oc_policy_weights &w, int64_t idx) {
    const int f = (int)(4 * ((idx >> 8) & 7) + (idx & 3)), lane = (int)((idx >> 2) & 63);
    const int m = f >> 4, s = f & 15, a = lane & 31;
    return a < L.n_out ? w.head_w[a * OC_POLICY_HID + 32 * m + oc_mfma_row(s, lane >> 5)] : 0.f;
}
// head bias, idx over [2 h][4 reg]: output a = reg + 4 h
OC_HD inline float oc_policy_pack_bh(const oc_policy_layout &L, const oc_policy_weights &w, int64_t idx) {
    const int a = (int)(idx & 3) + 4 * (int)(idx >> 2);
    return a < L.n_out ? w.head_b[a] : 0.f;
}

// Element e of the whole image in units of 4 bytes (the header and conv digits are assembled from bytes).
OC_HD inline uint32_t oc_policy_pack_word(const oc_policy_layout &L, const oc_policy_weights &w, int64_t word) {
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
    if (b < L.off_fc1) cv.f = oc_policy_pack_conv_sb(L, w, (b - L.off_conv_sb) / 4);
    else if (b < L.off_b1) cv.f = oc_policy_pack_fc1(L, w, (b - L.off_fc1) / 4);
    else if (b < L.off_fc2) cv.f = oc_policy_pack_hbias(w.fc1_b, (b - L.off_b1) / 4);
    else if (b < L.off_b2) cv.f = oc_policy_pack_fc2(w, (b - L.off_fc2) / 4);
    else if (b < L.off_head) cv.f = oc_policy_pack_hbias(w.fc2_b, (b - L.off_b2) / 4);
    else if (b < L.off_bh) cv.f = oc_policy_pack_head(L, w, (b - L.off_head) / 4);
    else if (b < L.off_bh + 32) cv.f = oc_policy_pack_bh(L, w, (b - L.off_bh) / 4);
    else return 0u;
    return cv.u;
}

// Uniform in (0, 1) for the Gumbel-max action draw: a pure function of (seed, counter, row, action), so a test
// can re-derive every sampled action from the returned log-probabilities.
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
