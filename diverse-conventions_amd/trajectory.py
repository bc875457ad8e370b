"""Time-major rollout storage that the step kernel fills directly.

The reference runner copies every observation three times after the env step -- `vobs.obs.clone()`, the
implicit int8->float32 cast into `turn_obs`, and `SharedReplayBuffer.chooseinsert` into the float32
`(T+1, N, agents, W, H, C)` buffer (train/MAPPO/main_player.py:245,229; utils/shared_buffer.py:115-148):
more HBM traffic than the env kernel itself, at 4x the bytes.  Here the kernel's destination for step t IS
slot t+1 of an int8 trajectory tensor; nothing is copied and the dtype stays int8 until a consumer casts.
"""
import torch

from .engine import OvercookedEngine


class TrajectoryBuffer:
    """obs int8 [T+1, P, N, W, H, C]; actions int32 [T, P, N]; reward int32 [T, P, N]; done int32 [T, N];
    optionally the compact native state records int32 [T+1, N, rec/4] (for `TrajectoryExchange`)."""

    def __init__(self, engine: OvercookedEngine, length: int, store_states: bool = False):
        self.engine, self.T = engine, int(length)
        n, dev = engine.num_kitchens, engine.device
        P = self.P = engine.num_players
        self.obs = torch.zeros((self.T + 1, P, n, engine.width, engine.height, engine.channels), dtype=torch.int8, device=dev)
        self.actions = torch.zeros((self.T, P, n), dtype=torch.int32, device=dev)
        self.reward = torch.zeros((self.T, P, n), dtype=torch.int32, device=dev)
        self.done = torch.zeros((self.T, n), dtype=torch.int32, device=dev)
        self.records = (torch.zeros((self.T + 1, n, engine.record_bytes // 4), dtype=torch.int32, device=dev)
                        if store_states else None)
        self.t = 0
        self.start()

    def start(self):
        """Slot 0 := observation of the engine's current states (episodes keep running, like n_reset)."""
        self.t = 0
        self.engine.observe(self.obs[0])
        if self.records is not None:
            self.engine.export_states(out=self.records[0])
        return self.obs[0]

    def step(self, actions):
        """actions [P, N] or [P, N, 1] (any integer dtype) -> (obs[t+1], reward[t], done[t]) views."""
        if self.t >= self.T:
            raise IndexError("trajectory is full: call after_update() or start()")
        t = self.t
        self.actions[t].copy_(actions.reshape(self.P, -1))
        self.engine.step(self.actions[t], obs=self.obs[t + 1], reward=self.reward[t], done=self.done[t])
        if self.records is not None:
            self.engine.export_states(out=self.records[t + 1])
        self.t = t + 1
        return self.obs[t + 1], self.reward[t], self.done[t]

    def rollout(self, actions):
        """Fill the whole buffer from a pre-drawn action tape [T, P, N] in ONE launch (oc_rollout_fused: the
        kitchens' state stays on chip between steps; 9.5 vs 13.6 us per step at 65 536 kitchens)."""
        if self.records is not None:
            raise ValueError("rollout() does not record states; use step()")
        self.start()
        self.actions.copy_(actions.reshape(self.T, self.P, -1))
        self.engine.rollout_fused(self.T, actions=self.actions, obs=self.obs[1:], reward=self.reward, done=self.done)
        self.t = self.T
        return self

    def after_update(self):
        """Last slot becomes the first of the next segment (reference shared_buffer.py:222-231)."""
        self.obs[0].copy_(self.obs[self.T])
        if self.records is not None:
            self.records[0].copy_(self.records[self.T])
        self.t = 0

    def episode_returns(self):
        """Mean return of the episodes that ENDED inside the buffer, per kitchen -> (mean f32 [N], count int64 [N]); NaN where
        none ended.  What the reference's runners accumulate as `running_score` and push into `scores` at every done
        (train/XD/xd_player.py:224-230); feed it to `sharding.segment_scores`.  Assumes slot 0 starts an episode."""
        r = self.reward[:self.t, 0].to(torch.float32)                 # the shared reward (sum over players)
        d = self.done[:self.t].to(torch.float32)
        upto_last_done = (r.cumsum(0) * d).amax(0)                     # rewards are never negative
        count = d.sum(0)
        return upto_last_done / count, count.to(torch.int64)

    def agent_view(self, agent: int):
        """Observations of one player as the reference buffer indexes them: [T+1, N, W, H, C] (int8 view)."""
        return self.obs[:, agent]
