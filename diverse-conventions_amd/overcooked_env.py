"""OvercookedMadrona: the reference's batched Overcooked environment class, served by the HIP engine.

Same constructor signature, attributes and return values as the reference class of that name
(envs/overcooked2_env.py:27-131), so `train/env_utils.py:generate_env` and the CoMeDi runners drop
in unchanged.  What differs is underneath: one fused HIP launch per step instead of the Madrona
task graph plus ~10 torch gather/scatter kernels.
"""
import numpy as np
import torch

from .engine import NUM_ACTIONS, OvercookedEngine
from .layouts import get_base_layout_params
from .vector_api import VectorMultiAgentEnv, VectorObservation

try:                                    # gym is optional: callers only read .shape / .n / class name
    from gym.spaces import Discrete, MultiBinary
except Exception:                       # pragma: no cover - exercised where gym is absent
    class MultiBinary:
        def __init__(self, n):
            self.n = n
            self.shape = tuple(int(x) for x in np.asarray(n).reshape(-1))
            self.dtype = np.int8

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.int64


class OvercookedMadrona(VectorMultiAgentEnv):
    def __init__(self, layout_name, num_envs, gpu_id, debug_compile=True, use_cpu=False, use_env_cpu=False,
                 ego_agent_idx=0, horizon=200, num_players=None, graph_step=False):
        """graph_step (not in the reference's signature; default off): `n_step` replays a one-launch HIP graph of the env step instead
        of enqueueing the launch through the C ABI each time -- measured, not adopted as the default (bench.py `drop_in_step`: a graph
        of one kernel costs as much to launch as the kernel; what a Python-driven step pays for is its Python)."""
        if use_cpu:
            raise RuntimeError("use_cpu=True (CPU simulation) is not provided: the MI355X engine has no CPU "
                               "execution mode; the CPU oracle under oracle/ is test infrastructure only")
        self.layout_name = layout_name
        self.base_layout_params = get_base_layout_params(layout_name, horizon, max_num_players=num_players)
        self.width = self.base_layout_params["width"]
        self.height = self.base_layout_params["height"]
        self.num_players = self.base_layout_params["num_players"]
        self.size = self.width * self.height
        self.horizon = horizon

        self.sim = OvercookedEngine(self.base_layout_params, num_envs, gpu_id)
        env_device = torch.device("cpu") if use_env_cpu else self.sim.device
        super().__init__(num_envs, device=env_device, ego_ind=ego_agent_idx, n_players=self.num_players)

        self.static_observations = self.sim.obs         # int8 [P, N, W, H, C]
        self.static_rewards = self.sim.reward           # int32 [P, N]
        self.static_dones = self.sim.done               # int32 [N]
        self.static_action_mask = torch.ones((num_envs, NUM_ACTIONS), dtype=torch.bool, device=env_device)
        self.static_active = torch.ones((num_envs,), dtype=torch.bool, device=env_device)
        self.channels = 5 * self.num_players + 10        # envs/overcooked2_env.py:49,97
        self.obs_size = self.size * self.channels
        self.state_size = self.obs_size
        self.infos = [{}] * self.num_envs
        # The simulator owns persistent output buffers and every step overwrites them in place (the reference's data-plane contract,
        # SURVEY 8(b)): the per-player views handed back by n_step are therefore the same objects every step when the env lives on
        # the GPU -- built once, not per step (a Python-driven step is host-bound: bench.py `drop_in_step`).
        self._on_device = env_device == self.sim.device
        self._views = None
        self._graph_step = bool(graph_step) and self._on_device
        self._graph, self._graph_actions_ptr = None, None

        self.observation_space = self._setup_observation_space()
        self.share_observation_space = self.observation_space
        self.action_space = Discrete(NUM_ACTIONS)
        self.n_reset()

    def _setup_observation_space(self):
        return MultiBinary(np.array([self.width, self.height, self.channels]))

    def to_torch(self, a):
        return a.to(self.device)

    def get_obs(self):
        if self._on_device:
            if self._views is None:
                self._views = [VectorObservation(self.static_active, self.static_observations[i], action_mask=self.static_action_mask)
                               for i in range(self.n_players)]
            return self._views
        obs = self.to_torch(self.static_observations)
        return [VectorObservation(self.static_active, obs[i], action_mask=self.static_action_mask)
                for i in range(self.n_players)]

    def n_step(self, actions):
        if self._graph_step and actions.device == self.sim.device and actions.dtype in (torch.int32, torch.int64) and actions.is_contiguous():
            if self._graph is None or self._graph_actions_ptr != actions.data_ptr():
                self._capture_step(actions)
            self._graph.replay()
        else:
            self.sim.step(actions)
        if self._on_device:
            return self.get_obs(), self.static_rewards, self.static_dones, self.infos
        return self.get_obs(), self.to_torch(self.static_rewards), self.to_torch(self.static_dones), self.infos

    def _capture_step(self, actions):
        """One-launch HIP graph of sim.step(actions) for this action buffer (VectorMultiAgentEnv stacks into a persistent one)."""
        self.sim.prepare_kernels()
        torch.cuda.synchronize(self.sim.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self.sim.step(actions)
        self._graph_actions_ptr = actions.data_ptr()

    def step(self, action):
        """VectorMultiAgentEnv.step (pantheonrl_extension/vectorenv.py:171-193) for the common case -- two players, everything on the
        GPU, integer action tensors -- without assembling the [P, N, 1] action tensor: the ego's and the partner's tensors go to the
        kernel as they are (oc_step_players; torch.stack was 7 of a step's 18 us).  Same calls on the partner (get_action with its
        observation, then update), same return values; `self._actions` is only refreshed on the general path."""
        if self._on_device and self.n_players == 2 and not self._graph_step and self._obs:
            other = 1 - self.ego_ind
            pact = self._current(0).get_action(self._obs[other])
            if torch.is_tensor(action) and torch.is_tensor(pact):
                pair = [action, pact] if self.ego_ind == 0 else [pact, action]
                if self.sim.step_players(pair):
                    self._obs = self.get_obs()
                    rews, done = self.static_rewards, self.static_dones
                    self._update_players(rews, done)
                    return self._obs[self.ego_ind], rews[self.ego_ind], done, self.infos
            # the general path, with the partner's action already drawn (an agent must not be asked twice for one step)
            per_player = [action, pact] if self.ego_ind == 0 else [pact, action]
            if self._actions is None:
                self._actions = torch.stack(per_player)
            else:
                torch.stack(per_player, out=self._actions)
            self._obs, rews, done, info = self.n_step(self._actions)
            self._update_players(rews, done)
            return self._obs[self.ego_ind], rews[self.ego_ind], done, info
        return super().step(action)

    def n_reset(self):
        # like the reference (envs/overcooked2_env.py:127-128) this re-reads the observation of the
        # running episodes; kitchens only reset inside step() when they reach the horizon
        return self.get_obs()

    def close(self, **kwargs):
        self.sim.close()
