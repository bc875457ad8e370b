"""Convention-paired rollout step: every policy of both sides in one fused MFMA launch + the fused env step, optionally replayed
as one HIP graph ("HIP-graph capture of the whole step").  (The PyTorch batched-GEMM twin of this class, the
comparison baseline, lives in tools/torch_policies.py since round 6.)

Kitchens are laid out in the reference's pairing segments [SP | XP0_i | XP1_i] (train/XD/serial.py:27); the ego
side of segment g is played by policy `ego[g]`, the partner side by `partner[g]` (xd_player.py:76-77,94).
"""
import torch

from .engine import OvercookedEngine
from .sharding import pairing_segments, segment_policy_indices


class FusedPairedRollout:
    """The paired rollout with every policy of the step on the fused MFMA kernel (`oc_policy_forward`): ONE policy launch
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
        self.steps_done = 0                                               # host-side draw counter of the eager path
        self.graph = None

    def _step(self):
        # graph capture: the draw counter lives on the device so that every replay samples afresh
        self.plan.forward(self.engine.obs, raw=self.values, actions=self.actions, action_logp=self.logp,
                          deterministic=self.deterministic, seed=self.seed, counter=0, device_counter=self.draw)
        self.draw.add_(1)
        self.engine.step(self.actions)

    def _native(self, T, in_place, obs, actions, logp, values, reward, done, records=None):
        import ctypes
        from . import native
        from .fused_policy import _ptr, _shape, _stream
        e, nets = self.engine, self.nets
        native.check(native.lib().oc_policy_rollout(
            e._handle, ctypes.byref(_shape(nets.W, nets.H, 6, nets.digits)), _ptr(nets.image), nets.n_nets, _ptr(self.plan.tiles),
            self.plan.n_tiles, int(T), 1 if in_place else 0, _ptr(obs), _ptr(actions), _ptr(logp), _ptr(values), _ptr(reward), _ptr(done),
            _ptr(records), 0 if self.deterministic else 1, int(self.seed) & 0xFFFFFFFF, self.steps_done & 0xFFFFFFFF, None, _stream(e.device)))
        self.steps_done += int(T)

    def step(self, steps: int = 1):
        """`steps` env steps for all segments (policy launch + env launch each, enqueued from C in one call)
        -> (obs, reward, done) engine buffers (overwritten every step)."""
        if self.graph is not None:
            for _ in range(steps):
                self.graph.replay()
        else:
            e = self.engine
            self._native(steps, True, e.obs, self.actions, self.logp, self.values, e.reward, e.done)
        return self.engine.obs, self.engine.reward, self.engine.done

    def capture(self, warmup: int = 3):
        self.engine.prepare_kernels()                         # nothing may compile or load a module inside the capture
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
        traj.require_time_major_obs("collect()")
        if traj.engine is not self.engine:
            raise ValueError("the trajectory belongs to another engine")
        traj.start()
        for name, t in (("logp", logp), ("values", values)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(traj.actions.shape)):
                raise ValueError(f"{name} must be a contiguous float32 tensor of shape {tuple(traj.actions.shape)}")
        if values is not None and self.values is None:
            raise ValueError("this rollout was built without critics")
        # (a trajectory built with store_states=True gets slot t+1 of its compact records from step t's env launch)
        self._native(traj.T, False, traj.obs, traj.actions, logp, values, traj.reward, traj.done, traj.records)
        traj.t = traj.T
        return traj
