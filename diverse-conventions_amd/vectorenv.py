"""Ego-view vector environment base class: `reset()/step()` for the ego agent on top of an
`n_reset()/n_step()` for all players, partners supplied as VectorAgents.

Own implementation of the surface in the reference's pantheonrl_extension/vectorenv.py:26-255
(constructor arguments, partner bookkeeping, resample policies, exceptions and return shapes are the
same, so the CoMeDi runners -- train/MAPPO/main_player.py:211-277, train/XD/xd_player.py:178-230 --
call it unchanged).
"""
from abc import ABC, abstractmethod
from typing import List, Optional

import numpy as np
import torch

from .vectoragent import VectorAgent


class PlayerException(Exception):
    """Raised when the players of the environment are set up incorrectly."""


class VectorMultiAgentEnv(ABC):
    def __init__(self, num_envs: int, device: torch.device, ego_ind: int = 0, n_players: int = 2,
                 resample_policy: str = "default", partners: Optional[List[List[VectorAgent]]] = None):
        self.num_envs = num_envs
        self.device = device
        self.ego_ind = ego_ind
        self.n_players = n_players
        if partners is not None:
            if len(partners) != n_players - 1:
                raise PlayerException("The number of partners needs to equal the number of non-ego players")
            for plist in partners:
                if not isinstance(plist, list) or not plist:
                    raise PlayerException("Sublist for each partner must be nonempty list")
        self.partners = partners or [[] for _ in range(n_players - 1)]
        self.partnerids = [0] * (n_players - 1)
        self._obs = tuple()
        self._actions = None
        self.set_resample_policy(resample_policy)

    # ------------------------------------------------------------------ partners
    def getDummyEnv(self, player_num: int):
        return self

    def _get_partner_num(self, player_num: int) -> int:
        if player_num == self.ego_ind:
            raise PlayerException("Ego agent is not set by the environment")
        return player_num - 1 if player_num > self.ego_ind else player_num

    def add_partner_agent(self, agent: VectorAgent, player_num: int = 1) -> None:
        self.partners[self._get_partner_num(player_num)].append(agent)

    def set_partnerid(self, agent_id: int, player_num: int = 1) -> None:
        partner_num = self._get_partner_num(player_num)
        assert 0 <= agent_id < len(self.partners[partner_num])
        self.partnerids[partner_num] = agent_id

    def resample_random(self) -> None:
        self.partnerids = [np.random.randint(len(plist)) for plist in self.partners]

    def resample_round_robin(self) -> None:
        self.partnerids = [(self.partnerids[0] + 1) % len(self.partners[0])]

    def set_resample_policy(self, resample_policy: str) -> None:
        if resample_policy == "default":
            resample_policy = "robin" if self.n_players == 2 else "random"
        if resample_policy == "robin" and self.n_players != 2:
            raise PlayerException("Cannot do round robin resampling for >2 players")
        if resample_policy == "robin":
            self.resample_partner = self.resample_round_robin
        elif resample_policy == "random":
            self.resample_partner = self.resample_random
        else:
            raise PlayerException(f"Invalid resampling policy: {resample_policy}")

    # ------------------------------------------------------------------ ego view
    def _get_actions(self, obs, ego_act=None):
        actions = []
        for player, ob in zip(range(self.n_players), obs):
            if player == self.ego_ind:
                actions.append(ego_act)
            else:
                p = self._get_partner_num(player)
                actions.append(self.partners[p][self.partnerids[p]].get_action(ob))
        if self._actions is None:
            self._actions = torch.stack(actions)
        else:
            torch.stack(actions, out=self._actions)
        return self._actions

    def _update_players(self, rews, done):
        for i in range(self.n_players - 1):
            playernum = i + (0 if i < self.ego_ind else 1)
            self.partners[i][self.partnerids[i]].update(rews[playernum], done)

    def step(self, action: torch.Tensor):
        """Ego action [N, 1] -> (ego VectorObservation, ego reward [N], done [N], info)."""
        acts = self._get_actions(self._obs, action)
        self._obs, rews, done, info = self.n_step(acts)
        self._update_players(rews, done)
        return self._obs[self.ego_ind], rews[self.ego_ind], done, info

    def reset(self):
        """Resample the partner and return the ego agent's current observation."""
        self.resample_partner()
        self._obs = self.n_reset()
        return self._obs[self.ego_ind]

    # ------------------------------------------------------------------ all players
    @abstractmethod
    def n_step(self, actions: torch.Tensor):
        """actions [P, N, 1] -> ([VectorObservation] * P, rewards [P, N], done [N], info)."""

    @abstractmethod
    def n_reset(self):
        """-> [VectorObservation] * P for the current states."""

    def close(self, **kwargs):
        pass
