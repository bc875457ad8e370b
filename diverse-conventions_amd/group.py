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
    def step(self, actions, obs=None, reward=None, done=None):
        """actions: one int32/int64 `[P, N_i]` (or `[P, N_i, 1]`) tensor per engine, all of one dtype.  Outputs go to
        each engine's own buffers unless explicit per-engine destination lists are given.  One launch, no host sync."""
        prepared = [e._actions(a) for e, a in zip(self.engines, actions)]
        if len(prepared) != len(self.engines):
            raise ValueError("one action tensor per engine")
        dtypes = {dt for _, dt in prepared}
        if len(dtypes) != 1:
            raise ValueError("the action tensors of a group step must share one dtype")
        acts = [a for a, _ in prepared]
        tabs = [self._own[i] if given is None else _table(given) for i, given in enumerate((obs, reward, done))]
        native.check(self._L.oc_group_step(self._handle, _table(acts), dtypes.pop(), tabs[0], tabs[1], tabs[2], self._stream()))
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
