"""numpy float32 restatement of the reference's GAE (train/MAPPO/utils/shared_buffer.py:248-304, use_gae
branches) and of the advantage subtraction (train/MAPPO/r_mappo.py:174-177).  TEST INFRASTRUCTURE ONLY.
Parity: pinned by tests/golden/gae_cases.npz (outputs of the reference itself, bit for bit)."""
import numpy as np

f32 = np.float32


def gae_returns(rewards, value_preds, masks, bad_masks=None, gamma=0.99, gae_lambda=0.95, denormalize=False,
                vn_mean=0.0, vn_std=1.0, proper_time_limits=False):
    """rewards [T, ...], value_preds/masks/bad_masks [T+1, ...] (value_preds[T] = next_value).
    -> returns [T, ...], advantages [T, ...]"""
    T = rewards.shape[0]
    g, gl = f32(gamma), f32(gamma * gae_lambda)                  # Python multiplies the two floats in double first

    def dn(v):                                                   # valuenorm.py:81-83
        return (v * f32(vn_std) + f32(vn_mean)).astype(f32) if denormalize else v

    returns = np.zeros_like(rewards, dtype=f32)
    gae = np.zeros_like(rewards[0], dtype=f32)
    for t in reversed(range(T)):
        v0, v1 = dn(value_preds[t]), dn(value_preds[t + 1])
        delta = rewards[t] + g * v1 * masks[t + 1] - v0          # :262-264 / :291-293
        if proper_time_limits:
            gae = delta + (gl * gae * masks[t + 1] if denormalize else gl * masks[t + 1] * gae)   # :265 / :272
            gae = gae * bad_masks[t + 1]                         # :266 / :273
        else:
            gae = delta + gl * masks[t + 1] * gae                # :294 / :299
        returns[t] = gae + v0
    adv = returns - (dn(value_preds[:-1]))                       # r_mappo.py:175-177
    return returns, adv.astype(f32)
