#!/usr/bin/env python3
"""Golden vectors for the mixed-play bookkeeping: the reference's own `XDPlayer.collect_mp_episode` /
`next_mp_step` (train/XD/xd_player.py:232-356), `MixedAgent.get_action` / `update` (train/partner_agents.py:153-244),
`VectorMultiAgentEnv.step` (pantheonrl_extension/vectorenv.py:146-193) and `SharedReplayBuffer.diaginsert` /
`partinsert` (train/MAPPO/utils/shared_buffer.py:150-220), imported in place from /root/reference and driven over
a counting environment with marker policies.  Every recorded quantity says which wall step, which env and which
policy produced it, so the file pins (a) which policy acts on which env at which step for given uniform draws and
(b) where every step of every env lands in the mixed-play buffer.  Stores inputs/outputs only.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mixed.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class Discrete(_Space):
    pass


_stub("gym", Env=object, Space=_Space, spaces=_stub("gym.spaces", Space=_Space, MultiBinary=_Space, Discrete=Discrete, Box=_Space))
import torch.utils  # noqa: E402,F401
_stub("torch.utils.tensorboard", SummaryWriter=object)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, "/root/reference/train")
from XD.xd_player import XDPlayer                                    # noqa: E402
from partner_agents import MixedAgent                                # noqa: E402
from MAPPO.utils.shared_buffer import SharedReplayBuffer             # noqa: E402
from pantheonrl_extension.vectorenv import VectorMultiAgentEnv       # noqa: E402
from pantheonrl_extension.vectorobservation import VectorObservation  # noqa: E402

OBS = (1, 1, 4)


class CountingEnv(VectorMultiAgentEnv):
    """obs of player p in env e after `t` steps = [t, e, p, steps since the last reset]; every env is done after
    `horizon` steps and resets inside that step (the Overcooked wrappers' rule); reward = 100 t + e."""

    def __init__(self, n, horizon):
        super().__init__(n, torch.device("cpu"))
        self.horizon, self.t, self.since = horizon, 0, 0
        self.observation_space = self.share_observation_space = _Space(OBS)
        self.action_space = Discrete(n=6)
        self.joint_actions = []

    def _obs_now(self):
        out = []
        for p in range(2):
            o = torch.zeros((self.num_envs, *OBS))
            o[:, 0, 0, 0], o[:, 0, 0, 1], o[:, 0, 0, 2], o[:, 0, 0, 3] = self.t, torch.arange(self.num_envs), p, self.since
            out.append(VectorObservation(torch.ones(self.num_envs, dtype=torch.bool), o, None, torch.ones((self.num_envs, 6))))
        return out

    def n_reset(self):
        return self._obs_now()

    def n_step(self, acts):
        self.joint_actions.append(acts.clone().numpy())
        rew = 100.0 * self.t + torch.arange(self.num_envs, dtype=torch.float)
        self.t += 1
        self.since += 1
        done = torch.full((self.num_envs,), int(self.since >= self.horizon), dtype=torch.int32)
        if self.since >= self.horizon:
            self.since = 0
        return self._obs_now(), [rew, rew], done, [{}] * self.num_envs


class MarkerActor:
    def __init__(self, ident):
        self.ident = ident

    def __call__(self, obs, rnn, masks, avail, deterministic=False):
        n = obs.shape[0]
        return torch.full((n, 1), float(self.ident)), -(self.ident + 1) - 0.001 * obs[:, 0, 0, 1:2], rnn


class MarkerCritic:
    def __init__(self, ident):
        self.ident = ident

    def __call__(self, cent_obs, rnn, masks):
        return 1000.0 * self.ident + cent_obs[:, 0, 0, 0:1], rnn


class MarkerPolicy:
    """stands for MCPolicy: actor 0 with critic 1; `mp_critic` is critic 2"""
    actor, critic, mp_critic = MarkerActor(0), MarkerCritic(1), MarkerCritic(2)

    def get_actions(self, cent_obs, obs, rnn, rnn_c, masks, avail=None, deterministic=False):
        a, lp, rnn = self.actor(obs, rnn, masks, avail)
        v, rnn_c = self.critic(cent_obs, rnn_c, masks)
        return v, a, lp, rnn, rnn_c


class Player(XDPlayer):
    def __init__(self, L, seed):
        self.episode_length, self.ego_id, self.device = L, 0, torch.device("cpu")
        self.envs_mp = CountingEnv(L - 1, L)
        args = types.SimpleNamespace(episode_length=L, n_rollout_threads=L - 1, hidden_size=2, recurrent_N=1, gamma=0.99,
                                     gae_lambda=0.95, use_gae=True, use_popart=False, use_valuenorm=False,
                                     use_proper_time_limits=False)
        self.mp_buf = SharedReplayBuffer(args, 2, _Space(OBS), _Space(OBS), Discrete(n=6), self.device, L - 1)
        self.policy = MarkerPolicy()
        self.agent_set, self.best_i = [MarkerActor(7)], 0           # the frozen convention chosen as mixed-play partner
        self.trainer = types.SimpleNamespace(prep_rollout=lambda: None)
        n, b = L - 1, self.mp_buf
        z = lambda t: torch.zeros((n, *t.shape[2:]))
        self.turn_mp_obs, self.turn_mp_share_obs = z(b.obs), z(b.share_obs)
        self.turn_mp_available_actions, self.turn_mp_values = z(b.available_actions), z(b.value_preds)
        self.turn_mp_actions, self.turn_mp_action_log_probs = z(b.actions), z(b.action_log_probs)
        self.turn_mp_rnn_states = z(b.rnn_states)
        self.turn_mp_rnn_states_critic = torch.zeros_like(self.turn_mp_rnn_states)
        self.turn_mp_masks = torch.ones((n, *b.masks.shape[2:]))
        self.turn_mp_active_masks, self.turn_mp_bad_masks = torch.ones_like(self.turn_mp_masks), torch.ones_like(self.turn_mp_masks)
        self.turn_mp_rewards = z(b.rewards)
        # XDPlayer.__init__ installs a MixedAgent; collect_mp_episode replaces it (xd_player.py:80-81, 235-237)
        self.envs_mp.add_partner_agent(MixedAgent(self, 1, [self.policy, self.policy], L))
        vobs = self.envs_mp.reset()                                  # warmup(), xd_player.py:376-385
        self.use_mp_obs, self.use_mp_share_obs = vobs.obs.clone(), vobs.state.clone()
        self.use_mp_available_actions = vobs.action_mask.clone()
        self.running_mp_score = torch.zeros(n)


def run(L, seed):
    torch.manual_seed(seed)
    draws = torch.rand((2 * L, 2, L - 1))          # reference order: per step, ego draw (next_mp_step) then partner draw (MixedAgent)
    torch.manual_seed(seed)
    pl = Player(L, seed)
    pl.collect_mp_episode()
    b = pl.mp_buf
    out = dict(L=np.int32(L), draws=draws.numpy(), joint_actions=np.stack(pl.envs_mp.joint_actions),
               mp_scores=np.asarray(pl.mp_scores, dtype=np.float32))
    for name in ("obs", "share_obs", "actions", "action_log_probs", "value_preds", "rewards", "masks", "active_masks",
                 "available_actions", "rnn_states"):
        out[name] = getattr(b, name).numpy()
    return out


if __name__ == "__main__":
    out = {}
    for i, (L, seed) in enumerate([(6, 3), (9, 4)]):
        for k, v in run(L, seed).items():
            out[f"c{i}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "mixed_play_cases.npz"), n_cases=np.int32(2), **out)
    print("wrote mixed_play_cases.npz", os.path.getsize(os.path.join(HERE, "mixed_play_cases.npz")) // 1024, "KiB")
    z = out
    print("joint actions of case 0 (step, player, env):\n", z["c0_joint_actions"][..., 0].astype(int).transpose(1, 0, 2))
    print("buffer actions agent 0 of case 0 [t, env]:\n", z["c0_actions"][:, :, 0, 0].astype(int))
    print("buffer obs[...,0] (wall step) agent 0 of case 0 [t, env]:\n", z["c0_obs"][:, :, 0, 0, 0, 0].astype(int))
