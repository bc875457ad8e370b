#!/usr/bin/env python3
"""Golden vectors for the self-play / cross-play collection: the reference's own
`XDPlayer.collect_episode` / `next_step` (train/XD/xd_player.py:83-149, 178-230), `CentralizedMultiAgent.get_action` / `update`
(train/partner_agents.py:97-152), `rMAPPOWrapper` (train/MAPPO/rMAPPOPolicy.py:129-132), `VectorMultiAgentEnv.step`
(pantheonrl_extension/vectorenv.py:146-193) and `SharedReplayBuffer.chooseinsert` (train/MAPPO/utils/shared_buffer.py:115-148),
imported in place from /root/reference and driven over a counting environment with marker policies for k = 1, 3, 7 frozen
conventions.  Every recorded quantity says which env row, which wall step and which network produced it, so the file pins
  (a) which ACTOR acts and which CRITIC values each env row on each side at every step (the segment -> policy map),
  (b) which env rows each of the 2k+1 buffers' `chooseinsert` receives (the segment slicing),
  (c) the score bookkeeping: per-segment episode returns, `xp_scores`, `best_i`.
Stores inputs / outputs only.  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pairing.py
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
from partner_agents import CentralizedMultiAgent                     # noqa: E402
from MAPPO.rMAPPOPolicy import rMAPPOWrapper                         # noqa: E402
from MAPPO.utils.shared_buffer import SharedReplayBuffer             # noqa: E402
from pantheonrl_extension.vectorenv import VectorMultiAgentEnv       # noqa: E402
from pantheonrl_extension.vectorobservation import VectorObservation  # noqa: E402

OBS = (1, 1, 4)
NEW_CRITIC, XP0_CRITIC, XP1_CRITIC = 1, 10, 20                       # critic idents: the new policy's, xp_critic0[i] = 10 + i, xp_critic1[i] = 20 + i


def reward_table(k, seed):
    """reward of a joint step by (ego actor ident, partner actor ident): small integers, so that which pairing scores best
    depends on the table and not on the segment order"""
    return np.random.default_rng(seed).integers(0, 9, size=(k + 1, k + 1)).astype(np.float32)


class CountingEnv(VectorMultiAgentEnv):
    """obs of player p in env e after `t` steps = [t, e, p, steps since the last reset]; every env is done after `horizon` steps
    and resets inside that step (the Overcooked wrappers' rule); reward = table[ego action, partner action] + e / 1024
    (the marker actors play their own ident, so the reward says who played)."""

    def __init__(self, n, horizon, table):
        super().__init__(n, torch.device("cpu"))
        self.horizon, self.t, self.since, self.table = horizon, 0, 0, torch.from_numpy(table)
        self.observation_space = self.share_observation_space = _Space(OBS)
        self.action_space = Discrete(n=6)
        self.joint_actions, self.rewards = [], []

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
        a = acts.clone().numpy()
        self.joint_actions.append(a)
        rew = self.table[acts[0, :, 0].long(), acts[1, :, 0].long()] + torch.arange(self.num_envs, dtype=torch.float) / 1024.0
        self.rewards.append(rew.numpy().copy())
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
        return torch.full((n, 1), float(self.ident)), -(self.ident + 1) - obs[:, 0, 0, 1:2] / 1024.0, rnn


class MarkerCritic:
    def __init__(self, ident):
        self.ident = ident

    def __call__(self, cent_obs, rnn, masks):
        return 1000.0 * self.ident + cent_obs[:, 0, 0, 0:1], rnn


class MarkerPolicy:
    """stands for MCPolicy (train/XD/MCPolicy.py:7-54): actor 0 = the convention being trained, its critic (ident 1) and the
    cross-play critics xp_critic0[i] (ident 10 + i), xp_critic1[i] (ident 20 + i)"""

    def __init__(self, k):
        self.actor, self.critic = MarkerActor(0), MarkerCritic(NEW_CRITIC)
        self.xp_critic0 = [MarkerCritic(XP0_CRITIC + i) for i in range(k)]
        self.xp_critic1 = [MarkerCritic(XP1_CRITIC + i) for i in range(k)]

    def get_actions(self, cent_obs, obs, rnn, rnn_c, masks, avail=None, deterministic=False):
        a, lp, rnn = self.actor(obs, rnn, masks, avail)
        v, rnn_c = self.critic(cent_obs, rnn_c, masks)
        return v, a, lp, rnn, rnn_c


class Player(XDPlayer):
    """XDPlayer with the constructor's wiring done by hand (xd_player.py:24-81): no networks, no trainer, no mixed play"""

    def __init__(self, k, threads, L, horizon, table):
        n = threads * (2 * k + 1)                                    # serial.py:27
        self.episode_length, self.ego_id, self.device = L, 0, torch.device("cpu")
        self.envs = CountingEnv(n, horizon, table)
        self.all_args = types.SimpleNamespace(episode_length=L, n_rollout_threads=threads, hidden_size=2, recurrent_N=1, gamma=0.99,
                                              gae_lambda=0.95, use_gae=True, use_popart=False, use_valuenorm=False,
                                              use_proper_time_limits=False)
        buf = lambda: SharedReplayBuffer(self.all_args, 2, _Space(OBS), _Space(OBS), Discrete(n=6), self.device)     # serial.py:16-19
        self.sp_buf, self.xp_buf0, self.xp_buf1 = buf(), [buf() for _ in range(k)], [buf() for _ in range(k)]
        self.buffer = self.sp_buf
        self.policy = MarkerPolicy(k)
        self.agent_set = [MarkerActor(1 + i) for i in range(k)]      # frozen convention i plays ident 1 + i
        self.xp_weight, self.mp_weight = 0.25, 0.0
        self.trainer = types.SimpleNamespace(prep_rollout=lambda: None)
        b = self.sp_buf                                              # the turn_* scratch of MainPlayer.setup_data (main_player.py:263-277), all env rows
        z = lambda t: torch.zeros((n, *t.shape[2:]))
        self.turn_obs, self.turn_share_obs = z(b.obs), z(b.share_obs)
        self.turn_available_actions, self.turn_values = z(b.available_actions), z(b.value_preds)
        self.turn_actions, self.turn_action_log_probs = z(b.actions), z(b.action_log_probs)
        self.turn_rnn_states = z(b.rnn_states)
        self.turn_rnn_states_critic = torch.zeros_like(self.turn_rnn_states)
        self.turn_masks = torch.ones((n, *b.masks.shape[2:]))
        self.turn_active_masks, self.turn_bad_masks = torch.ones_like(self.turn_masks), torch.ones_like(self.turn_masks)
        self.turn_rewards = z(b.rewards)
        # xd_player.py:72-77: the partner side is played by [new] + wrapped conventions (xp_critic1) + [new] * k
        wrapped = [rMAPPOWrapper(x, self.policy.xp_critic1[i]) for i, x in enumerate(self.agent_set)]
        self.envs.add_partner_agent(CentralizedMultiAgent(self, 1 - self.ego_id, [self.policy] + wrapped + [self.policy] * k, threads))
        vobs = self.envs.reset()                                     # warmup(), main_player.py:279-287
        self.use_obs, self.use_share_obs = vobs.obs.clone(), vobs.state.clone()
        self.use_available_actions = vobs.action_mask.clone()


def run(k, threads, L, horizon, seed):
    table = reward_table(k, seed)
    pl = Player(k, threads, L, horizon, table)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):                  # ("best i is ..." of xd_player.py:148)
        pl.collect_episode()
    fields = ("obs", "share_obs", "actions", "action_log_probs", "value_preds", "rewards", "masks", "active_masks")
    bufs = [pl.sp_buf] + pl.xp_buf0 + pl.xp_buf1                     # segment order [SP | XP0_i | XP1_i]
    out = dict(k=np.int32(k), threads=np.int32(threads), L=np.int32(L), horizon=np.int32(horizon), xp_weight=np.float32(pl.xp_weight),
               table=table, joint_actions=np.stack(pl.envs.joint_actions)[..., 0].astype(np.int32),      # [L, side, env]
               env_rewards=np.stack(pl.envs.rewards),                                                     # [L, env]
               best_i=np.int32(-1 if pl.best_i is None else pl.best_i))
    for f in fields:
        out["buf_" + f] = np.stack([getattr(bb, f) if isinstance(getattr(bb, f), np.ndarray) else getattr(bb, f).numpy() for bb in bufs])   # [2k+1, ...]
    # scores: one list per segment (all the same length here: every env of a segment finishes at the same steps)
    out["scores"] = np.asarray(pl.scores, dtype=np.float32)                                               # [2k+1, episodes * threads]
    out["sp_scores"] = np.asarray(pl.sp_scores, dtype=np.float32)
    out["xp_scores"] = np.asarray(pl.xp_scores, dtype=np.float32).reshape(2, k, -1)                       # [side, i, episodes * threads]
    return out


if __name__ == "__main__":
    out = {}
    cases = [(1, 3, 6, 6, 11), (3, 2, 7, 3, 12), (7, 2, 5, 5, 13), (3, 4, 9, 4, 14)]      # (k, threads, L, horizon, seed)
    for i, c in enumerate(cases):
        for kk, v in run(*c).items():
            out[f"c{i}_{kk}"] = v
    path = os.path.join(HERE, "pairing_cases.npz")
    np.savez_compressed(path, n_cases=np.int32(len(cases)), **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    for i, c in enumerate(cases):
        print(f"case {i} (k, threads, L, horizon, seed) = {c}: best_i {out[f'c{i}_best_i']}, ego actors of step 0 {out[f'c{i}_joint_actions'][0, 0].tolist()}, "
              f"partner actors {out[f'c{i}_joint_actions'][0, 1].tolist()}")
        print("   ego critics (value // 1000) of step 0 per buffer:", (out[f"c{i}_buf_value_preds"][:, 0, :, 0, 0] // 1000).astype(int).tolist())
        print("   partner critics:", (out[f"c{i}_buf_value_preds"][:, 0, :, 1, 0] // 1000).astype(int).tolist())
