#!/usr/bin/env python3
"""Golden vectors for the update-side data path: the reference's advantage normalisation
(train/XD/xd.py:233-245, same lines as train/MAPPO/r_mappo.py:174-181) and its minibatch generators
(SharedReplayBuffer.feed_forward_generator / partial_ff_generator, train/MAPPO/utils/shared_buffer.py:306-429),
imported in place from /root/reference, run on seeded buffers.  Stores inputs, the permutation torch.randperm drew
and the yielded minibatches only.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_minibatch.py
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
from XD.xd import XD                                          # noqa: E402

FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns", "masks",
          "active_masks", "action_log_probs", "adv_targ", "available_actions")


class Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class Discrete(Space):
    pass


def run(T, N, A, obs_shape, seed, use_valuenorm, inactive, num_mini_batch):
    g = torch.Generator().manual_seed(seed)
    args = types.SimpleNamespace(episode_length=T, n_rollout_threads=N, hidden_size=4, recurrent_N=1, gamma=0.99,
                                 gae_lambda=0.95, use_gae=True, use_popart=False, use_valuenorm=use_valuenorm,
                                 use_proper_time_limits=False)
    buf = SharedReplayBuffer(args, A, Space(obs_shape), Space(obs_shape), Discrete(n=6), torch.device("cpu"))
    buf.obs.copy_(torch.randint(0, 21, buf.obs.shape, generator=g).float() * (torch.rand(buf.obs.shape, generator=g) < 0.3))
    buf.share_obs.copy_(buf.obs)                              # Overcooked: state is the observation (overcooked2_env.py:97-98)
    buf.rnn_states.copy_(torch.randn(buf.rnn_states.shape, generator=g))
    buf.rnn_states_critic.copy_(torch.randn(buf.rnn_states.shape, generator=g))
    buf.actions.copy_(torch.randint(0, 6, buf.actions.shape, generator=g).float())
    buf.action_log_probs.copy_(-torch.rand(buf.actions.shape, generator=g) * 2)
    buf.value_preds.copy_(torch.randn(buf.value_preds.shape, generator=g) * 1.3 + 0.2)
    buf.returns.copy_(torch.randn(buf.returns.shape, generator=g) * 2.1 + 0.5)
    buf.masks.copy_((torch.rand(buf.masks.shape, generator=g) > 0.1).float())
    if inactive:
        buf.active_masks.copy_((torch.rand(buf.masks.shape, generator=g) > 0.25).float())
    vn, mean, std = None, 0.0, 1.0
    if use_valuenorm:
        vn = ValueNorm(1, device=torch.device("cpu"))
        vn.update(torch.randn((64, 1), generator=g) * 3.0 + 1.0)
        m, v = vn.running_mean_var()
        mean, std = float(m[0]), float(torch.sqrt(v)[0])
    trainer = types.SimpleNamespace(_use_popart=False, _use_valuenorm=use_valuenorm, value_normalizer=vn)
    adv = XD.calc_advantanges(trainer, buf)                   # normalised advantages, [T, N, A, 1]
    dv = vn.denormalize(buf.value_preds[:-1]) if vn is not None else buf.value_preds[:-1]
    out = dict(T=np.int32(T), N=np.int32(N), A=np.int32(A), num_mini_batch=np.int32(num_mini_batch),
               obs=buf.obs.numpy().astype(np.int8), rnn_states=buf.rnn_states.numpy(),
               rnn_states_critic=buf.rnn_states_critic.numpy(), actions=buf.actions.numpy(),
               action_log_probs=buf.action_log_probs.numpy(), value_preds=buf.value_preds.numpy(),
               returns=buf.returns.numpy(), masks=buf.masks.numpy(), active_masks=buf.active_masks.numpy(),
               raw_advantages=(buf.returns[:-1] - dv).numpy(), advantages=adv.numpy(),
               use_valuenorm=np.int32(use_valuenorm), vn_mean=np.float32(mean), vn_std=np.float32(std))
    if torch.isnan(adv).any():                                # the minibatches would only repeat the NaNs
        adv = buf.returns[:-1] - dv
        out["generator_advantages"] = adv.numpy()
    for name, gen, batch in (("ff", lambda: buf.feed_forward_generator(adv, num_mini_batch), T * N * A),
                             ("p0", lambda: buf.partial_ff_generator(adv, 0, num_mini_batch), T * N),
                             ("p1", lambda: buf.partial_ff_generator(adv, 1, num_mini_batch), T * N)):
        torch.manual_seed(seed + 1)
        out[f"{name}_perm"] = torch.randperm(batch).numpy()
        torch.manual_seed(seed + 1)
        batches = list(gen())
        for k, f in enumerate(FIELDS):
            out[f"{name}_{f}"] = np.stack([b[k].numpy() for b in batches])
    for name in ("ff", "p0", "p1"):
        assert np.array_equal(out[f"{name}_share_obs"], out[f"{name}_obs"])
        del out[f"{name}_share_obs"]                          # identical to obs by construction
        out[f"{name}_obs"] = out[f"{name}_obs"].astype(np.int8)
    return out


if __name__ == "__main__":
    out = {}
    specs = [(6, 5, 2, (3, 2, 20), 11, True, False, 3),       # 120-byte rows (not a multiple of 16)
             (5, 7, 2, (4, 2, 20), 12, False, False, 4),      # 160-byte rows, batch not divisible by the minibatch count
             (4, 6, 2, (5, 4, 20), 13, True, True, 2)]        # inactive agents: the reference's std over NaNs
    for i, s in enumerate(specs):
        for k, v in run(*s).items():
            out[f"c{i}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "minibatch_cases.npz"), n_cases=np.int32(len(specs)), **out)
    print("wrote minibatch_cases.npz", os.path.getsize(os.path.join(HERE, "minibatch_cases.npz")) // 1024, "KiB")
