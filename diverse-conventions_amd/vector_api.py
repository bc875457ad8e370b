"""The vector-env surface the CoMeDi runners program against, in one module.

Public names, constructor arguments, attributes and return shapes follow the reference's
`pantheonrl_extension/{vectorenv,vectoragent,vectorobservation}.py` (VectorMultiAgentEnv :26-255,
VectorAgent :9-29, RandomVectorAgent :32-41, VectorObservation :5-32) so that
`train/MAPPO/main_player.py:211-277` and `train/XD/xd_player.py:178-230` call it unchanged; the
implementation is this repo's own.

Roles: the *ego* player's actions come from the caller of `step()`; every other player slot owns a
roster of partner agents, one of which is current and is asked for actions each step.
"""
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch


class PlayerException(Exception):
    """The player / partner configuration of an environment is inconsistent."""


class VectorObservation:
    """What one player sees in N kitchens at once.

    active       bool [N]           which kitchens want an action from this player
    obs          [N, *obs_shape]
    state        [N, *state_shape]  defaults to `obs` (the reference's rule)
    action_mask  bool [N, A] or None (None = everything allowed)
    """
    __slots__ = ("active", "obs", "state", "action_mask")

    def __init__(self, active, obs, state=None, action_mask=None):
        self.active, self.obs, self.action_mask = active, obs, action_mask
        self.state = obs if state is None else state

    def __repr__(self):
        shape = lambda t: None if t is None else tuple(t.shape)
        return f"VectorObservation(obs={shape(self.obs)}, state={shape(self.state)}, mask={shape(self.action_mask)})"


class VectorAgent:
    """Partner protocol: `get_action(obs, record=True) -> [N, 1]`, `update(rewards [N], dones [N])`."""

    def get_action(self, obs: VectorObservation, record: bool = True) -> torch.Tensor:
        raise NotImplementedError

    def update(self, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        raise NotImplementedError


class RandomVectorAgent(VectorAgent):
    """Plays whatever `sampler()` returns; learns nothing."""

    def __init__(self, sampler: Callable[[], torch.Tensor]):
        self.sampler = sampler

    def get_action(self, obs, record=True):
        return self.sampler()

    def update(self, rewards, dones):
        pass


class VectorMultiAgentEnv:
    """Ego-view wrapper: subclasses provide `n_step(actions [P, N, 1])` and `n_reset()` for all players,
    this class turns them into `step(ego_action)` / `reset()` for one player plus partner bookkeeping."""

    RESAMPLERS = ("default", "robin", "random")

    def __init__(self, num_envs: int, device: torch.device, ego_ind: int = 0, n_players: int = 2,
                 resample_policy: str = "default", partners: Optional[List[List[VectorAgent]]] = None):
        self.num_envs, self.device = num_envs, device
        self.ego_ind, self.n_players = ego_ind, n_players
        slots = n_players - 1
        if partners is None:
            partners = [[] for _ in range(slots)]
        else:
            if len(partners) != slots:
                raise PlayerException("The number of partners needs to equal the number of non-ego players")
            if any(not isinstance(roster, list) or not roster for roster in partners):
                raise PlayerException("Sublist for each partner must be nonempty list")
        self.partners = partners            # partners[slot] = roster of agents that may play that slot
        self.partnerids = [0] * slots       # partnerids[slot] = index of the current agent in its roster
        self._obs: Sequence[VectorObservation] = ()
        self._actions = None                # [P, N, 1], reused between steps
        self.set_resample_policy(resample_policy)

    # -- slots ------------------------------------------------------------------------------------
    def _get_partner_num(self, player_num: int) -> int:
        """Roster slot of a (non-ego) player index."""
        if player_num == self.ego_ind:
            raise PlayerException("Ego agent is not set by the environment")
        return player_num - (player_num > self.ego_ind)

    def _player_of_slot(self, slot: int) -> int:
        return slot + (slot >= self.ego_ind)

    def _current(self, slot: int) -> VectorAgent:
        return self.partners[slot][self.partnerids[slot]]

    def getDummyEnv(self, player_num: int):
        return self

    def add_partner_agent(self, agent: VectorAgent, player_num: int = 1) -> None:
        self.partners[self._get_partner_num(player_num)].append(agent)

    def set_partnerid(self, agent_id: int, player_num: int = 1) -> None:
        slot = self._get_partner_num(player_num)
        if not 0 <= agent_id < len(self.partners[slot]):
            raise AssertionError(f"agent_id {agent_id} outside the roster of player {player_num}")
        self.partnerids[slot] = agent_id

    # -- who plays next episode -------------------------------------------------------------------
    def resample_random(self) -> None:
        # numpy's global generator, like the reference, so `np.random.seed` reproduces the same partner draws
        self.partnerids = [int(np.random.randint(len(roster))) for roster in self.partners]

    def resample_round_robin(self) -> None:
        self.partnerids = [(self.partnerids[0] + 1) % len(self.partners[0])]

    def set_resample_policy(self, resample_policy: str) -> None:
        if resample_policy not in self.RESAMPLERS:
            raise PlayerException(f"Invalid resampling policy: {resample_policy}")
        if resample_policy == "default":
            resample_policy = "robin" if self.n_players == 2 else "random"
        if resample_policy == "robin" and self.n_players != 2:
            raise PlayerException("Cannot do round robin resampling for >2 players")
        self.resample_partner = self.resample_round_robin if resample_policy == "robin" else self.resample_random

    # -- ego view ---------------------------------------------------------------------------------
    def _get_actions(self, obs, ego_act=None):
        per_player = [ego_act if p == self.ego_ind else self._current(self._get_partner_num(p)).get_action(obs[p])
                      for p in range(self.n_players)]
        if self._actions is None:
            self._actions = torch.stack(per_player)
        else:
            torch.stack(per_player, out=self._actions)
        return self._actions

    def _update_players(self, rews, done):
        for slot in range(self.n_players - 1):
            self._current(slot).update(rews[self._player_of_slot(slot)], done)

    def step(self, action: torch.Tensor):
        """ego action [N, 1] -> (ego VectorObservation, ego reward [N], done [N], info)"""
        self._obs, rews, done, info = self.n_step(self._get_actions(self._obs, action))
        self._update_players(rews, done)
        return self._obs[self.ego_ind], rews[self.ego_ind], done, info

    def reset(self):
        """Pick the next partner(s) and return the ego player's current observation."""
        self.resample_partner()
        self._obs = self.n_reset()
        return self._obs[self.ego_ind]

    # -- all players (subclass responsibility) ----------------------------------------------------
    def n_step(self, actions: torch.Tensor):
        """actions [P, N, 1] -> ([VectorObservation] * P, rewards [P, N], done [N], info)"""
        raise NotImplementedError

    def n_reset(self):
        """-> [VectorObservation] * P for the current states"""
        raise NotImplementedError

    def close(self, **kwargs):
        pass
