#!/usr/bin/env python3
"""Golden vectors for segment-batched policy inference: the reference's R_Actor
(train/MAPPO/r_actor_critic.py:11-76 with CNNBase utils/cnn.py:12-60 and ACTLayer utils/act.py) imported in
place, three seeded random actors (hidden 64, as train/train_xp.sh uses), real int8 observations of `simple`.
Stores weights, inputs, action logits, deterministic actions and their log-probs.  Build container only."""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/train")
from MAPPO.r_actor_critic import R_Actor, R_Critic      # noqa: E402


class Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class Discrete(Space):
    pass


if __name__ == "__main__":
    args = types.SimpleNamespace(hidden_size=64, gain=0.01, use_orthogonal=True, use_policy_active_masks=True,
                                 use_naive_recurrent_policy=False, use_recurrent_policy=False, recurrent_N=1,
                                 use_ReLU=True, use_feature_normalization=True, stacked_frames=1, layer_N=2, use_popart=False)
    obs = np.load(os.path.join(HERE, "guided_simple.npz"))["obs"]          # [T, P, W, H, C] int8
    obs = obs[::3].reshape(-1, 5, 4, 20)[:240]
    out = {"obs": obs}
    for k in range(3):
        torch.manual_seed(100 + k)
        actor = R_Actor(args, Space((5, 4, 20)), Discrete(n=6))
        with torch.no_grad():
            for p in actor.parameters():                  # default init has zero biases and tiny action gains
                p.add_(torch.randn_like(p) * 0.05)
            feats = actor.base(torch.from_numpy(obs).float())
            logits = actor.act.action_out(feats).logits   # normalised log-probabilities
            act, logp, _ = actor(torch.from_numpy(obs).float(), torch.zeros(len(obs), 1, 64), torch.ones(len(obs), 1),
                                 deterministic=True)
        for name, t in actor.state_dict().items():
            out[f"a{k}.{name}"] = t.numpy()
        out[f"a{k}.logits"] = logits.numpy()
        out[f"a{k}.det_action"] = act.numpy()
        out[f"a{k}.det_logp"] = logp.numpy()
    for k in range(2):                                    # critics: same trunk, scalar value head (r_actor_critic.py:139-199)
        torch.manual_seed(200 + k)
        critic = R_Critic(args, Space((5, 4, 20)))
        with torch.no_grad():
            for p in critic.parameters():
                p.add_(torch.randn_like(p) * 0.05)
            values, _ = critic(torch.from_numpy(obs).float(), torch.zeros(len(obs), 1, 64), torch.ones(len(obs), 1))
        for name, t in critic.state_dict().items():
            out[f"c{k}.{name}"] = t.numpy()
        out[f"c{k}.values"] = values.numpy()
    # a second grid shape (unident_s, 9x5 -> 7x3 conv positions) pins the Flatten order (o, x', y') where W-2 != H-2 != 2
    obs2 = np.load(os.path.join(HERE, "guided_unident_s.npz"))["obs"][::5].reshape(-1, 9, 5, 20)[:96]
    out2 = {"obs": obs2}
    torch.manual_seed(300)
    actor = R_Actor(args, Space((9, 5, 20)), Discrete(n=6))
    critic = R_Critic(args, Space((9, 5, 20)))
    with torch.no_grad():
        for p in list(actor.parameters()) + list(critic.parameters()):
            p.add_(torch.randn_like(p) * 0.05)
        x2 = torch.from_numpy(obs2).float()
        out2["a0.logits"] = actor.act.action_out(actor.base(x2)).logits.numpy()
        out2["c0.values"] = critic(x2, torch.zeros(len(obs2), 1, 64), torch.ones(len(obs2), 1))[0].numpy()
    for name, t in actor.state_dict().items():
        out2[f"a0.{name}"] = t.numpy()
    for name, t in critic.state_dict().items():
        out2[f"c0.{name}"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "policy_cases_unident_s.npz"), **out2)
    np.savez_compressed(os.path.join(HERE, "policy_cases.npz"), **out)
    print("wrote policy_cases.npz", os.path.getsize(os.path.join(HERE, "policy_cases.npz")) // 1024, "KiB",
          [k for k in out if k.startswith("a0.")])
