"""EngineGroup: several engines (one per layout) stepped by ONE kernel launch.

The reference trains and evaluates on its five layouts side by side with one simulator object per layout
(`generate_env`, train/env_utils.py:26; the evaluation sweep of train/test_all.sh:8; BASELINE configs 3 and 5), i.e.
five `sim.step()` calls per joint step.  Here the engines stay separate objects -- each keeps its own buffers and its
`reset()/step()` surface -- and the group adds the joint step: one launch of a module that holds one specialised
kernel body per distinct layout (`oc_group_step`, include/overcooked_hip.h).
"""
import ctypes
from typing import Optional, Sequence

import torch

from . import native
from .fused_policy import PolicyPlan
from .engine import OvercookedEngine

_vp = ctypes.c_void_p


def _table(tensors):
    return (_vp * len(tensors))(*[_vp(t.data_ptr()) for t in tensors])


class EngineGroup:
    def __init__(self, engines: Sequence[OvercookedEngine]):
        self._handle = None
        self.engines = list(engines)
        if not self.engines:
            raise ValueError("a group needs at least one engine")
        self.device = self.engines[0].device
        self._L = native.lib()
        handle = _vp()
        native.check(self._L.oc_group_create((_vp * len(self.engines))(*[e._handle for e in self.engines]), len(self.engines),
                                             ctypes.byref(handle)))
        self._handle = handle
        self._own = tuple(_table([getattr(e, k) for e in self.engines]) for k in ("obs", "reward", "done"))

    @classmethod
    def from_layouts(cls, layouts, num_kitchens, device_index=0, horizon=200):
        """One engine of `num_kitchens` kitchens (an int, or one count per layout) for every layout name."""
        from .layouts import get_base_layout_params
        counts = [num_kitchens] * len(layouts) if isinstance(num_kitchens, int) else list(num_kitchens)
        return cls([OvercookedEngine(get_base_layout_params(name, horizon), n, device_index) for name, n in zip(layouts, counts)])

    # ------------------------------------------------------------------ lifetime
    def close(self, close_engines: bool = False):
        if self._handle is not None:
            torch.cuda.synchronize(self.device)
            self._L.oc_group_destroy(self._handle)
            self._handle = None
        if close_engines:
            for e in self.engines:
                e.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.engines)

    def _stream(self):
        return _vp(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def build_info(self) -> str:
        return self._L.oc_group_build_info(self._handle).decode()

    # ------------------------------------------------------------------ hot path
    def step(self, actions, obs=None, reward=None, done=None, records=None):
        """actions: one int32/int64 `[P, N_i]` (or `[P, N_i, 1]`) tensor per engine, all of one dtype.  Outputs go to
        each engine's own buffers unless explicit per-engine destination lists are given.  records: optional list of int32
        `[N_i, rec_i/4]` tensors (entries may be None) that receive the post-step state records.  One launch, no host sync."""
        prepared = [e._actions(a) for e, a in zip(self.engines, actions)]
        if len(prepared) != len(self.engines):
            raise ValueError("one action tensor per engine")
        dtypes = {dt for _, dt in prepared}
        if len(dtypes) != 1:
            raise ValueError("the action tensors of a group step must share one dtype")
        acts = [a for a, _ in prepared]
        tabs = [self._own[i] if given is None else _table(given) for i, given in enumerate((obs, reward, done))]
        rec_tab = None
        if records is not None:
            ptrs = [e._records_arg(r)[0] for e, r in zip(self.engines, records)]
            rec_tab = (ctypes.c_void_p * len(ptrs))(*[p.value if p is not None else None for p in ptrs])
        native.check(self._L.oc_group_step(self._handle, _table(acts), dtypes.pop(), tabs[0], tabs[1], tabs[2], rec_tab, self._stream()))
        pick = lambda given, name: [getattr(e, name) for e in self.engines] if given is None else list(given)
        return pick(obs, "obs"), pick(reward, "reward"), pick(done, "done")

    def observe(self, obs=None):
        tab = self._own[0] if obs is None else _table(obs)
        native.check(self._L.oc_group_observe(self._handle, tab, self._stream()))
        return [e.obs for e in self.engines] if obs is None else list(obs)

    def reset_all(self):
        for e in self.engines:
            native.check(self._L.oc_reset_all(e._handle, None, e._stream()))
        return self.observe()

    def rollout_timed(self, actions, steps: int) -> float:
        """`steps` joint steps enqueued from C, engine i cycling through its tape `actions[i]` `[T_act, P, N_i]` (int32);
        returns the elapsed milliseconds measured with HIP events on the launch stream."""
        t_act = actions[0].shape[0]
        if any(a.shape[0] != t_act or a.dtype != torch.int32 or not a.is_contiguous() for a in actions):
            raise ValueError("action tapes must be contiguous int32 [T_act, P, N_i] with one T_act")
        ms = ctypes.c_float(0.0)
        native.check(self._L.oc_group_rollout_timed(self._handle, _table(actions), native.OC_ACT_I32, t_act, int(steps), self._own[0],
                                                    self._own[1], self._own[2], self._stream(), ctypes.byref(ms)))
        return ms.value

    def set_tuning(self, kitchens_per_wave: Optional[Sequence[int]] = None, passes_per_flush: int = 0, waves_per_cu: int = 0):
        kpw = None if kitchens_per_wave is None else (ctypes.c_int32 * len(self.engines))(*[int(k) for k in kitchens_per_wave])
        native.check(self._L.oc_group_set_tuning(self._handle, kpw, passes_per_flush, waves_per_cu))


class GroupPairedRollout:
    """Convention-paired rollout over the engines of an `EngineGroup` with the policies in the loop: what `FusedPairedRollout`
    does for one engine, for all of them at once, every step enqueued from C (`oc_group_policy_rollout`).  The reference does
    this as five simulators and 5 x 2(2k+1) sequential `get_actions` calls per joint step (train/env_utils.py:26,
    train/XD/xd_player.py:190-207).  Kitchens of different engines never interact, so the engines need not march in step:

    partition="one_launch" (= "auto", the default): TWO launches per joint step on the current stream -- one policy launch over all
        engines' rows (`oc_policy_forward_multi`: one tile table balanced by block cost; per-layout launches cannot share a CU,
        so at 8192 kitchens per layout they leave half the SIMDs idle) and one env launch (`oc_group_step`): config 3's shape
        (8 conventions, 15 segments, 5 x 8190 kitchens) 95 us per joint step, 155 us with critics; config 5's (32 conventions, 63
        segments, critics) 191 us -- its 130-kitchen segments run as "wave tiles": one-wavefront workgroups that keep their whole
        network in L2 / registers, four different networks per CU (profiles/r02_bench_group_policies.log).
    partition=False: the engines' policy launches one after the other, then ONE env launch for all engines: 144 / 229 us.
    Results are identical in both modes.  (Every engine on its own stream -- 128 / 176 us -- and on CU-masked streams -- 328 us --
    were measured and not adopted; they live in tools/exp_group_streams.py over include/overcooked_experiments.h.)

    nets[i]: PackedNetworks of engine i's grid; threads[i] kitchens per pairing segment (engine i holds threads[i] * (2k+1));
    critic_nets[i][side][g] as in FusedPairedRollout (or None)."""

    def __init__(self, group: EngineGroup, nets, threads, n_conventions: int, critic_nets=None, deterministic: bool = False, seed: int = 0,
                 partition="auto"):
        from .fused_policy import TILE_ACT, _shape
        from .sharding import segment_policy_indices
        from .sharding import pairing_segments
        self.group, self.nets, self.deterministic, self.seed = group, list(nets), deterministic, int(seed)
        n_eng = len(group)
        self.threads = [threads] * n_eng if isinstance(threads, int) else list(threads)
        self.segments, rows_all, work = [], [], []
        for i, (e, nt, th) in enumerate(zip(group.engines, self.nets, self.threads)):
            if e.num_players != 2:
                raise native.OvercookedNativeError(-2, f"engine {i} has {e.num_players} players: the fused policy kernel reads 2-player "
                                                       "observations, and the paired rollout's action / reward buffers are [2, N]")
            segs = pairing_segments(th, n_conventions)
            g, n = len(segs), e.num_kitchens
            if n != g * th:
                raise ValueError(f"engine {i} must hold threads*(2k+1) = {g * th} kitchens")
            actors = [segment_policy_indices(segs, "ego"), segment_policy_indices(segs, "partner")]
            rows = []
            for side in range(2):
                for s in range(g):
                    rows.append((side * n + s * th, th, actors[side][s], TILE_ACT))
                    if critic_nets is not None and critic_nets[i] is not None:
                        rows.append((side * n + s * th, th, critic_nets[i][side][s], 0))
            self.segments.append(segs)
            rows_all.append(rows)
            positions = (e.width - 2) * (e.height - 2)                     # MFMA + VALU cycles per 32-row block
            work.append(len(rows) * th / 32.0 * (positions * 4100.0 + 8500.0))
        self.has_critics = [critic_nets is not None and critic_nets[i] is not None for i in range(n_eng)]
        L = native.lib()
        self._L = L
        if partition == "auto":
            # one policy launch over all grids + one env launch: 95 / 155 us per joint step against 128 / 176 on five streams
            # (config 3's shape), 191 against 202 us with config 5's short segments, which the planner runs as wave tiles
            partition = "one_launch"
        if partition not in (False, None, "one_launch"):
            raise ValueError("partition must be 'auto', 'one_launch' or False (the per-engine-stream experiments: tools/exp_group_streams.py)")
        self.partition = partition or False
        self._one_launch = self.partition == "one_launch"
        if self._one_launch:
            from .fused_policy import MultiPolicyPlan
            self.multi_plan = MultiPolicyPlan(self.nets, [2 * e.num_kitchens for e in group.engines], rows_all)
        self._rows_all, self._work = rows_all, work
        self.plans = self._make_plans(rows_all)
        dev = group.device
        self.actions = [torch.zeros((2, e.num_kitchens), dtype=torch.int32, device=dev) for e in group.engines]
        self.logp = [torch.zeros((2, e.num_kitchens), dtype=torch.float32, device=dev) for e in group.engines]
        self.values = [torch.zeros((2, e.num_kitchens), dtype=torch.float32, device=dev) if c else None
                       for e, c in zip(group.engines, self.has_critics)]
        self._shapes = [_shape(nt.W, nt.H, 6, nt.digits) for nt in self.nets]
        self.steps_done = 0

    def _make_plans(self, rows_all):
        cus_total = torch.cuda.get_device_properties(self.group.device).multi_processor_count
        return [PolicyPlan(nt, 2 * e.num_kitchens, rows, rows_per_tile=-cus_total) for e, nt, rows in zip(self.group.engines, self.nets, rows_all)]

    def _slots(self, obs, actions, logp, values, reward, done, records=None):
        p = lambda t: _vp(t.data_ptr()) if t is not None else None
        slots = (native.OcGroupPolicySlot * len(self.group))()
        for i, (nt, plan) in enumerate(zip(self.nets, self.plans)):
            s = slots[i]
            s.shape = ctypes.pointer(self._shapes[i])
            s.d_packed, s.n_nets, s.d_tiles, s.n_tiles = p(nt.image), nt.n_nets, p(plan.tiles), plan.n_tiles
            s.d_obs, s.d_actions, s.d_action_logp, s.d_values = p(obs[i]), p(actions[i]), p(logp[i]), p(values[i])
            s.d_reward, s.d_done = p(reward[i]), p(done[i])
            s.d_records = p(records[i]) if records is not None else None
        return slots

    def _run(self, T, in_place, obs, actions, logp, values, reward, done, records=None):
        slots = self._slots(obs, actions, logp, values, reward, done, records)
        sample, seed, c0 = 0 if self.deterministic else 1, self.seed & 0xFFFFFFFF, self.steps_done & 0xFFFFFFFF
        if self._one_launch:
            native.check(self._L.oc_group_policy_rollout_multi(
                self.group._handle, slots, _vp(self.multi_plan.tiles.data_ptr()), self.multi_plan.n_tiles, self.multi_plan.tiles_per_workgroup,
                self.multi_plan.waves_per_workgroup, int(T), 1 if in_place else 0, sample, seed, c0, self.group._stream()))
        else:
            native.check(self._L.oc_group_policy_rollout(self.group._handle, slots, int(T), 1 if in_place else 0, sample, seed, c0,
                                                         self.group._stream()))
        self.steps_done += int(T)

    def step(self, steps: int = 1):
        """`steps` joint steps, outputs overwritten in place in every engine's own buffers."""
        es = self.group.engines
        self._run(steps, True, [e.obs for e in es], self.actions, self.logp, self.values, [e.reward for e in es], [e.done for e in es])
        return [e.obs for e in es], [e.reward for e in es], [e.done for e in es]

    def collect(self, trajs, logps=None, values=None):
        """Fill one TrajectoryBuffer per engine (same length T): step t reads slot t of each, writes actions into it and the env
        launch writes every engine's slot t+1.  logps / values: optional lists of float32 [T, 2, N_i] tensors."""
        T = trajs[0].T
        if any(t.T != T for t in trajs):
            raise ValueError("the trajectories of a group rollout must share one length")
        if len(trajs) != len(self.group.engines) or any(t.engine is not e for t, e in zip(trajs, self.group.engines)):
            raise ValueError("one trajectory per engine of the group, in the group's order")
        for t in trajs:
            t.require_time_major_obs("collect()")
        for t in trajs:
            t.start()
        none = [None] * len(trajs)
        # (trajectories built with store_states=True get their compact records from the env launches themselves)
        self._run(T, False, [t.obs for t in trajs], [t.actions for t in trajs], logps or none, values or none,
                  [t.reward for t in trajs], [t.done for t in trajs], [t.records for t in trajs])
        for t in trajs:
            t.t = T
        return trajs

    def close(self):
        pass
