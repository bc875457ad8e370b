#!/usr/bin/env python3
"""Numerics probe (CPU, numpy): the dense layers of the fused policy kernel as f16 TWO-piece products (3 MFMAs per tile:
hi*hi + hi*lo + lo*hi, operands split with round-to-nearest, power-of-two scales that keep the low pieces in f16's normal
range) against today's bf16 THREE-piece products (6 MFMAs per tile), both against float64 arithmetic on the f32 weights and
against the reference's own f32 outputs (tests/golden/policy_cases*.npz).  Conv: exact integers on 3 base-256 digits, as shipped."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.policy_oracle import reference_forward, split_bf16_trunc

C, O = 20, 32
QMAX = 126 * 256 * 256


def conv_units(sd, obs):
    """exact integer conv sums in units of each channel's scale, and the scales"""
    cw, cb = sd["base.cnn.cnn.0.weight"].astype(np.float32), sd["base.cnn.cnn.0.bias"].astype(np.float32)
    absmax = np.abs(cw).reshape(O, -1).max(1)
    scale = np.maximum((absmax.astype(np.float64) / QMAX), np.abs(cb).astype(np.float64) / 2**30).astype(np.float32)
    scale[scale <= 0] = 1.0
    q = np.rint(cw.astype(np.float64) / scale[:, None, None, None].astype(np.float64)).astype(np.int64)
    bu = np.rint(cb.astype(np.float64) / scale.astype(np.float64)).astype(np.int64)
    x = obs.astype(np.int64)
    B, W, H, _ = x.shape
    t = np.zeros((B, O, W - 2, H - 2), dtype=np.int64)
    for kx in range(3):
        for ky in range(3):
            t += np.einsum("bxyc,oc->boxy", x[:, kx:kx + W - 2, ky:ky + H - 2, :], q[:, :, kx, ky])
    return t + bu[None, :, None, None], scale


def mfma_chain(wpieces, xpieces, terms, bias, kchunk=16):
    """acc[n][b] = bias + sum over k-chunks (one MFMA each, per term): products exact, the K sum of one MFMA in f64, one rounding to f32 per MFMA"""
    n_out, K = wpieces[0].shape
    acc = np.tile(bias.astype(np.float32)[:, None], (1, xpieces[0].shape[1]))
    for k0 in range(0, K, kchunk):
        for (wi, xi) in terms:
            d = wpieces[wi][:, k0:k0 + kchunk].astype(np.float64) @ xpieces[xi][k0:k0 + kchunk].astype(np.float64)
            acc = (acc.astype(np.float64) + d).astype(np.float32)
    return acc


def split_f16(v):
    hi = v.astype(np.float16)
    r = (v.astype(np.float32) - hi.astype(np.float32)).astype(np.float32)
    lo = r.astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def split_bf16_rne3(w):
    out, r = [], w.astype(np.float32)
    for _ in range(3):
        u = r.view(np.uint32)
        b = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
        p = (b << 16).view(np.float32)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def pow2_for(maxabs, target=2.0**14):
    return 2.0 ** np.floor(np.log2(target / max(maxabs, 1e-30)))


def forward(sd, obs, head, scheme, B_act=16.0, K0=19):
    t, scale = conv_units(sd, obs)
    Bn = t.shape[0]
    P = t.shape[2] * t.shape[3]
    tf = t.astype(np.float32)                                        # v_cvt_f32_i32 (+ fma of the digits): one rounding
    w1 = sd["base.cnn.cnn.3.weight"].astype(np.float32)              # [64][O*P], column o*P + p
    b1, w2, b2 = sd["base.cnn.cnn.3.bias"].astype(np.float32), sd["base.cnn.cnn.5.weight"].astype(np.float32), sd["base.cnn.cnn.5.bias"].astype(np.float32)
    wh, bh = sd[head + ".weight"].astype(np.float32), sd[head + ".bias"].astype(np.float32)
    sc_col = np.repeat(scale, P)                                     # per fc1 input column
    if scheme == "bf16x3":
        T6 = ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0))
        x = np.maximum(tf, 0).reshape(Bn, -1).T                      # [K][B], units of scale
        w1s = (w1 * sc_col[None, :]).astype(np.float32)
        h1 = mfma_chain(split_bf16_rne3(w1s), list(split_bf16_trunc(x)), T6, b1)
        h2 = mfma_chain(split_bf16_rne3(w2), list(split_bf16_trunc(np.maximum(h1, 0))), T6, b2)
        return mfma_chain(split_bf16_rne3(wh), list(split_bf16_trunc(np.maximum(h2, 0))), T6, bh).T
    T3 = ((0, 0), (0, 1), (1, 0))
    clamp = lambda v: np.minimum(np.maximum(v, 0), np.float32(65504.0)).astype(np.float32)
    x = clamp(tf * np.float32(2.0 ** -K0)).reshape(Bn, -1).T         # 2^(23-K0) x true / absmax
    w1s = (w1.astype(np.float64) * sc_col[None, :].astype(np.float64) * 2.0 ** K0).astype(np.float32)
    A1 = pow2_for(np.abs(w1s).max())
    h1 = mfma_chain(list(split_f16((w1s * np.float32(A1)).astype(np.float32))), list(split_f16(x)), T3, (b1 * np.float32(A1)).astype(np.float32))
    x2 = clamp(h1 * np.float32(B_act / A1))
    A2 = pow2_for(np.abs(w2).max())
    h2 = mfma_chain(list(split_f16((w2 * np.float32(A2)).astype(np.float32))), list(split_f16(x2)), T3, (b2 * np.float32(A2 * B_act)).astype(np.float32))
    x3 = clamp(h2 * np.float32(1.0 / A2))
    A3 = pow2_for(np.abs(wh).max())
    out = mfma_chain(list(split_f16((wh * np.float32(A3)).astype(np.float32))), list(split_f16(x3)), T3, (bh * np.float32(A3 * B_act)).astype(np.float32))
    stats = dict(x1max=float(x.max()), x2max=float(x2.max()), x3max=float(x3.max()))
    return (out * np.float32(1.0 / (A3 * B_act))).astype(np.float32).T, stats


def f64_forward(sd, obs, head):
    x = obs.astype(np.float64)
    B, W, H, _ = x.shape
    cw = sd["base.cnn.cnn.0.weight"].astype(np.float64)
    conv = np.zeros((B, O, W - 2, H - 2))
    for kx in range(3):
        for ky in range(3):
            conv += np.einsum("bxyc,oc->boxy", x[:, kx:kx + W - 2, ky:ky + H - 2, :], cw[:, :, kx, ky])
    f = np.maximum(conv + sd["base.cnn.cnn.0.bias"].astype(np.float64)[None, :, None, None], 0).reshape(B, -1)
    f = np.maximum(f @ sd["base.cnn.cnn.3.weight"].T.astype(np.float64) + sd["base.cnn.cnn.3.bias"], 0)
    f = np.maximum(f @ sd["base.cnn.cnn.5.weight"].T.astype(np.float64) + sd["base.cnn.cnn.5.bias"], 0)
    return f @ sd[head + ".weight"].T.astype(np.float64) + sd[head + ".bias"]


def main():
    for fname in ("policy_cases.npz", "policy_cases_unident_s.npz"):
        z = np.load(os.path.join(ROOT, "tests", "golden", fname))
        obs = z["obs"]
        for pre, head in (("a0.", "act.action_out.linear"), ("a1.", "act.action_out.linear"), ("c0.", "v_out"), ("c1.", "v_out")):
            sd = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and ("weight" in k or "bias" in k)}
            if not sd:
                continue
            ref32 = reference_forward(sd, obs, head)
            ref64 = f64_forward(sd, obs, head)
            old = forward(sd, obs, head, "bf16x3")
            new, st = forward(sd, obs, head, "f16x2")
            mag = np.abs(ref64).max()
            e = lambda a: (np.abs(a - ref64).max(), np.abs(a - ref32).max())
            print(f"{fname} {pre} |out|max {mag:.2f}: f32 ref vs f64 {np.abs(ref32 - ref64).max():.2e}; bf16x3 vs f64 {e(old)[0]:.2e} vs ref {e(old)[1]:.2e}; "
                  f"f16x2 vs f64 {e(new)[0]:.2e} vs ref {e(new)[1]:.2e}; scaled activations max {st}")
    # stress: large random networks / inputs (logits to +-30), as tools/policy_accuracy.py does
    rng = np.random.default_rng(0)
    for (W, H), s in (((5, 4), 0.1), ((9, 5), 0.1), ((5, 4), 0.4), ((9, 5), 0.25)):
        P = (W - 2) * (H - 2)
        r = lambda *sh: (rng.standard_normal(sh) * s).astype(np.float32)
        sd = {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P), "base.cnn.cnn.3.bias": r(64),
              "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64), "h.weight": r(6, 64), "h.bias": r(6)}
        obs = (rng.random((256, W, H, 20)) < 0.15).astype(np.int8)
        obs[..., 16] = rng.integers(0, 21, size=obs.shape[:-1]) * (rng.random(obs.shape[:-1]) < 0.1)
        ref32, ref64 = reference_forward(sd, obs, "h"), f64_forward(sd, obs, "h")
        old = forward(sd, obs, "h", "bf16x3")
        new, st = forward(sd, obs, "h", "f16x2")
        print(f"random {W}x{H} sigma {s}: |out|max {np.abs(ref64).max():.1f}; f32 ref vs f64 {np.abs(ref32 - ref64).max():.2e}; bf16x3 {np.abs(old - ref64).max():.2e}; "
              f"f16x2 {np.abs(new - ref64).max():.2e}; scaled activations max {st}")


if __name__ == "__main__":
    main()
