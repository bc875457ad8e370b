#!/usr/bin/env python3
"""Golden vectors for the GAE op: runs the reference's SharedReplayBuffer.compute_returns
(train/MAPPO/utils/shared_buffer.py:248-304) and ValueNorm (train/MAPPO/utils/valuenorm.py) imported in
place from /root/reference on seeded inputs and stores inputs + outputs only.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gae.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/train")
from MAPPO.utils.shared_buffer import SharedReplayBuffer      # noqa: E402
from MAPPO.utils.valuenorm import ValueNorm                   # noqa: E402


class Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class Discrete(Space):
    pass


def run(T, N, A, seed, use_valuenorm, proper, gamma=0.99, lam=0.95):
    g = torch.Generator().manual_seed(seed)
    args = types.SimpleNamespace(episode_length=T, n_rollout_threads=N, hidden_size=8, recurrent_N=1, gamma=gamma,
                                 gae_lambda=lam, use_gae=True, use_popart=False, use_valuenorm=use_valuenorm,
                                 use_proper_time_limits=proper)
    buf = SharedReplayBuffer(args, A, Space((3, 2, 4)), Space((3, 2, 4)), Discrete(n=6), torch.device("cpu"))
    buf.rewards.copy_(torch.randint(0, 4, buf.rewards.shape, generator=g).float() * (torch.rand(buf.rewards.shape, generator=g) < 0.2) * 3.0
                      + (torch.rand(buf.rewards.shape, generator=g) < 0.02) * 20.0)
    buf.value_preds.copy_(torch.randn(buf.value_preds.shape, generator=g) * 1.7 + 0.3)
    buf.masks.copy_((torch.rand(buf.masks.shape, generator=g) > 0.03).float())
    buf.bad_masks.copy_((torch.rand(buf.bad_masks.shape, generator=g) > 0.05).float())
    next_value = torch.randn((N, A, 1), generator=g)
    vn, mean, std = None, 0.0, 1.0
    if use_valuenorm:
        vn = ValueNorm(1, device=torch.device("cpu"))
        for _ in range(3):
            vn.update(torch.randn((64, 1), generator=g) * 5.0 + 2.0)
        m, v = vn.running_mean_var()
        mean, std = float(m[0]), float(torch.sqrt(v)[0])
    values_in = buf.value_preds.clone()
    values_in[-1] = next_value
    buf.compute_returns(next_value, vn)
    dv = vn.denormalize(buf.value_preds[:-1]) if vn is not None else buf.value_preds[:-1]
    adv = buf.returns[:-1] - dv
    return dict(rewards=buf.rewards.numpy(), value_preds=values_in.numpy(), masks=buf.masks.numpy(),
                bad_masks=buf.bad_masks.numpy(), returns=buf.returns[:-1].numpy(), advantages=adv.numpy(),
                gamma=np.float64(gamma), gae_lambda=np.float64(lam), use_valuenorm=np.int32(use_valuenorm),
                proper_time_limits=np.int32(proper), vn_mean=np.float32(mean), vn_std=np.float32(std))


if __name__ == "__main__":
    out = {}
    i = 0
    for use_vn in (1, 0):
        for proper in (0, 1):
            for k, v in run(60, 37, 2, 100 + i, bool(use_vn), bool(proper)).items():
                out[f"c{i}_{k}"] = v
            i += 1
    for k, v in run(200, 5, 2, 7, True, False, gamma=0.995, lam=0.9).items():
        out[f"c{i}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "gae_cases.npz"), n_cases=np.int32(i + 1), **out)
    print("wrote gae_cases.npz", os.path.getsize(os.path.join(HERE, "gae_cases.npz")) // 1024, "KiB")
