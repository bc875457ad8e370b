#!/usr/bin/env python3
"""EXPERIMENT (measured, not adopted; include/overcooked_experiments.h): the five-layout rollout with every engine on its OWN
stream -- ordinary streams ("streams") or CU-masked ones whose shares follow each layout's policy work ("cus").

    python tools/exp_group_streams.py [streams|cus] [critics]

Config 3's shape (8 conventions, 15 segments, 5 x 8190 kitchens), profiles/r02_bench_group_policies.log: 128 / 176 us per joint
step on five streams and 328 us on CU-masked streams, against 95 / 155 us for the product path (ONE policy launch over all
grids + ONE env launch, GroupPairedRollout).  Results are identical in every mode; the self-check below compares the
collected trajectories with the product path's before timing.
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from diverse_conventions_amd import (CLASSIC_LAYOUTS, EngineGroup, GroupPairedRollout, PackedNetworks, TrajectoryBuffer,
                                     get_base_layout_params, native)
from diverse_conventions_amd.fused_policy import PolicyPlan

_vp = ctypes.c_void_p


class StreamedGroupPairedRollout(GroupPairedRollout):
    """GroupPairedRollout whose engines each run their whole rollout on a stream of their own
    (oc_exp_group_policy_rollout_streams); mode "cus": CU-masked streams (oc_exp_cu_partition_create)."""

    def __init__(self, group, nets, threads, n_conventions, mode="streams", **kw):
        super().__init__(group, nets, threads, n_conventions, partition=False, **kw)
        n = len(group)
        self.mode = mode
        self._streams = (_vp * n)()
        self._cus = (ctypes.c_int32 * n)()
        self._torch_streams = None
        if mode == "cus":
            shares = (ctypes.c_float * n)(*[w / sum(self._work) for w in self._work])
            native.check(self._L.oc_exp_cu_partition_create(group.device.index or 0, shares, n, self._streams, self._cus))
            self.plans = [PolicyPlan(nt, 2 * e.num_kitchens, rows, rows_per_tile=-int(self._cus[i]))
                          for i, (e, nt, rows) in enumerate(zip(group.engines, self.nets, self._rows_all))]
        elif mode == "streams":
            self._torch_streams = [torch.cuda.Stream(group.device) for _ in range(n)]
            for i, st in enumerate(self._torch_streams):
                self._streams[i] = st.cuda_stream
        else:
            raise ValueError("mode: 'streams' or 'cus'")

    @property
    def cu_shares(self):
        return [int(c) for c in self._cus] if self.mode == "cus" else None

    def _run(self, T, in_place, obs, actions, logp, values, reward, done):
        slots = self._slots(obs, actions, logp, values, reward, done)
        native.check(self._L.oc_exp_group_policy_rollout_streams(
            self.group._handle, slots, self._streams, self._cus if self.mode == "cus" else None, int(T), 1 if in_place else 0,
            0 if self.deterministic else 1, self.seed & 0xFFFFFFFF, self.steps_done & 0xFFFFFFFF, self.group._stream()))
        self.steps_done += int(T)

    def close(self):
        if self.mode == "cus" and self._streams is not None:
            torch.cuda.synchronize(self.group.device)
            self._L.oc_exp_cu_partition_destroy(self._streams, len(self.group))
            self._streams = None


def make_nets(names, k, critics, gen):
    nets, critic_of = [], []
    G = 2 * k + 1
    for nm in names:
        p = get_base_layout_params(nm, 200)
        W, H = p["width"], p["height"]
        P = (W - 2) * (H - 2)
        r = lambda *s: torch.randn(*s, generator=gen) * 0.1
        trunk = lambda: {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                         "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64)}
        actors = [dict(trunk(), **{"act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}) for _ in range(k + 1)]
        crit = [dict(trunk(), **{"v_out.weight": r(1, 64), "v_out.bias": r(1)}) for _ in range(2)] if critics else []
        nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=crit))
        critic_of.append([[k + 1 + (g % 2) for g in range(G)], [k + 1 + ((g + 1) % 2) for g in range(G)]] if critics else None)
    return nets, (critic_of if critics else None)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "streams"
    critics = len(sys.argv) > 2 and sys.argv[2] == "critics"
    k, n = 7, 8190
    threads = n // (2 * k + 1)
    gen = torch.Generator().manual_seed(0)
    nets, cn = make_nets(CLASSIC_LAYOUTS, k, critics, gen)
    # self-check on a short collection: same trajectories as the product path
    T = 6
    got = []
    for cls, kw in ((GroupPairedRollout, {}), (StreamedGroupPairedRollout, {"mode": mode})):
        grp = EngineGroup.from_layouts(CLASSIC_LAYOUTS, threads * (2 * k + 1), 0)
        ro = cls(grp, nets, threads, k, critic_nets=cn, seed=5, **kw)
        trajs = [TrajectoryBuffer(e, T) for e in grp.engines]
        ro.collect(trajs)
        torch.cuda.synchronize()
        got.append([(t.actions.clone(), t.obs.clone()) for t in trajs])
        if cls is StreamedGroupPairedRollout:
            ro.step(50)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ro.step(300)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 300 * 1e6
            print(f"{mode}{' + critics' if critics else ''}: {us:.1f} us per joint step, cu shares {ro.cu_shares}")
        ro.close()
        grp.close(close_engines=True)
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(*got))
    print("identical to the product path:", same)
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
