"""Convention-paired rollout step: batched policies for both sides + the fused env step, optionally replayed
as one HIP graph ("HIP-graph capture of the whole step").

Kitchens are laid out in the reference's pairing segments [SP | XP0_i | XP1_i] (train/XD/serial.py:27); the ego
side of segment g is played by policy `ego[g]`, the partner side by `partner[g]` (xd_player.py:76-77,94).
"""
import torch

from .engine import OvercookedEngine
from .policy import StackedActors, segment_policy_indices
from .sharding import pairing_segments


class PairedRollout:
    def __init__(self, engine: OvercookedEngine, actors: StackedActors, threads: int, n_conventions: int,
                 ego_id: int = 0, deterministic: bool = False):
        self.engine, self.actors, self.threads = engine, actors, threads
        self.segments = pairing_segments(threads, n_conventions)
        g = len(self.segments)
        if engine.num_kitchens != g * threads:
            raise ValueError(f"engine must hold threads*(2k+1) = {g * threads} kitchens")
        dev = engine.device
        self.ego_id = ego_id
        self.idx = [None, None]
        self.idx[ego_id] = torch.tensor(segment_policy_indices(self.segments, "ego"), device=dev)
        self.idx[1 - ego_id] = torch.tensor(segment_policy_indices(self.segments, "partner"), device=dev)
        self.selected = [actors.select(i) for i in self.idx]     # gathered once per pairing
        self.deterministic = deterministic
        self.actions = torch.zeros((2, engine.num_kitchens), dtype=torch.int64, device=dev)
        self.logp = torch.zeros((2, engine.num_kitchens), dtype=actors.dtype, device=dev)
        self.graph = None

    def _step(self):
        e = self.engine
        g = len(self.segments)
        for p in range(2):                                   # both sides, every segment, four GEMMs each
            obs = e.obs[p].view(g, self.threads, e.width, e.height, e.obs.shape[-1])
            a, lp = self.actors.act(obs, deterministic=self.deterministic, selected=self.selected[p])
            self.actions[p].copy_(a.view(-1))
            self.logp[p].copy_(lp.view(-1))
        e.step(self.actions)                                 # int64 actions are read as they are

    def step(self):
        """One env step for all segments -> (obs, reward, done) engine buffers (overwritten next step)."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step()
        return self.engine.obs, self.engine.reward, self.engine.done

    def capture(self, warmup: int = 3):
        """Record policy GEMMs + sampling + env kernel once; replay costs one graph launch per step."""
        s = torch.cuda.Stream(self.engine.device)
        s.wait_stream(torch.cuda.current_stream(self.engine.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream(self.engine.device).wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        return self
