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


class FusedPairedRollout:
    """PairedRollout with every policy of the step on the fused MFMA kernel (`oc_policy_forward`): ONE policy launch
    -- all conventions' actors, optionally their critics, both sides, every pairing segment -- plus ONE env launch
    per step, instead of the reference's 2(2k+1) `get_actions` calls (train/XD/xd_player.py:190-207,
    train/partner_agents.py:97-111).  The kernel reads the engine's int8 observations and writes the int32 actions
    `oc_step` reads; log-probabilities and values land in `[2, N]` float buffers (or a trajectory slot).

    actor_nets[side][g] / critic_nets[side][g]: network index in `nets` playing / valuing segment g on that side
    (defaults: actors from `pairing_segments`; no critics)."""

    def __init__(self, engine: OvercookedEngine, nets, threads: int, n_conventions: int, actor_nets=None, critic_nets=None,
                 ego_id: int = 0, deterministic: bool = False, seed: int = 0, rows_per_tile=None):
        from .fused_policy import TILE_ACT, PolicyPlan
        self.engine, self.nets, self.threads = engine, nets, threads
        self.segments = pairing_segments(threads, n_conventions)
        g, n = len(self.segments), engine.num_kitchens
        if n != g * threads:
            raise ValueError(f"engine must hold threads*(2k+1) = {g * threads} kitchens")
        if actor_nets is None:
            actor_nets = [None, None]
            actor_nets[ego_id] = segment_policy_indices(self.segments, "ego")
            actor_nets[1 - ego_id] = segment_policy_indices(self.segments, "partner")
        rows = []
        for side in range(2):
            for s in range(g):
                rows.append((side * n + s * threads, threads, actor_nets[side][s], TILE_ACT))
                if critic_nets is not None:
                    rows.append((side * n + s * threads, threads, critic_nets[side][s], 0))
        self.plan = PolicyPlan(nets, 2 * n, rows, rows_per_tile=rows_per_tile)
        dev = engine.device
        self.deterministic, self.seed = deterministic, seed
        self.actions = torch.zeros((2, n), dtype=torch.int32, device=dev)
        self.logp = torch.zeros((2, n), dtype=torch.float32, device=dev)
        self.values = torch.zeros((2, n), dtype=torch.float32, device=dev) if critic_nets is not None else None
        self.draw = torch.zeros(1, dtype=torch.int32, device=dev)        # device-side draw counter: graph replays differ
        self.graph = None

    def _policies(self, obs, actions, logp, values):
        self.plan.forward(obs, raw=values, actions=actions, action_logp=logp, deterministic=self.deterministic,
                          seed=self.seed, counter=0, device_counter=self.draw)
        self.draw.add_(1)

    def _step(self):
        self._policies(self.engine.obs, self.actions, self.logp, self.values)
        self.engine.step(self.actions)

    def step(self):
        """One env step for all segments -> (obs, reward, done) engine buffers (overwritten next step)."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step()
        return self.engine.obs, self.engine.reward, self.engine.done

    def capture(self, warmup: int = 3):
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

    def collect(self, traj, logp=None, values=None):
        """Fill a TrajectoryBuffer: step t reads `traj.obs[t]`, writes actions straight into `traj.actions[t]` and the
        env step writes `traj.obs[t+1]`, `traj.reward[t]`, `traj.done[t]` -- two launches per step, no copies.
        logp / values: optional float32 [T, 2, N] destinations."""
        traj.start()
        for t in range(traj.T):
            self._policies(traj.obs[t], traj.actions[t], None if logp is None else logp[t], None if values is None else values[t])
            self.engine.step(traj.actions[t], obs=traj.obs[t + 1], reward=traj.reward[t], done=traj.done[t])
        traj.t = traj.T
        return traj
