#!/usr/bin/env python3
"""Secondary measurements on one GPU for the BASELINE.json configs that are not the headline bench line,
plus the time-major-destination variant of the headline (observations written to a fresh slot every step).
Prints one JSON object per measurement."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import algorithmic_bytes
from diverse_conventions_amd import CLASSIC_LAYOUTS, OvercookedEngine, get_base_layout_params, pairing_segments


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def single(layout, n, steps=1000):
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, n, 0)
    actions = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
    eng.rollout(actions, 100)
    ms = min(eng.rollout_timed(actions, steps) for _ in range(3))
    us = ms * 1e3 / steps
    gbs = algorithmic_bytes(S) * n / us / 1e3
    out = {"what": "one launch per step, outputs overwritten in place", "layout": layout, "kitchens": n,
           "us_per_launch": round(us, 2), "agent_env_steps_per_s": round(2 * n / us * 1e6), "algorithmic_GBps": round(gbs, 1),
           "roofline_frac": round(gbs / 8000, 3), "kernel_build": eng.kernel_build_info}
    eng.close()
    return out


def time_major(layout, n, T=200):
    """Observations/rewards/dones of every step written into their own time-major slot (what a trajectory
    buffer wants): the destination is T x 52 MB, far beyond the 256 MB Infinity Cache."""
    params = get_base_layout_params(layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, n, 0)
    actions = torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32)
    obs = torch.empty((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
    rew = torch.empty((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.empty((T, n), dtype=torch.int32, device="cuda")
    eng.rollout(actions, T, obs=obs, reward=rew, done=done)
    sec = timed(lambda: eng.rollout(actions, T, obs=obs, reward=rew, done=done))
    us = sec * 1e6 / T
    gbs = algorithmic_bytes(S) * n / us / 1e3
    out = {"what": "one launch per step, outputs into time-major trajectory slots", "layout": layout, "kitchens": n, "T": T,
           "trajectory_GB": round(obs.numel() / 1e9, 2), "us_per_launch": round(us, 2),
           "agent_env_steps_per_s": round(2 * n / us * 1e6), "algorithmic_GBps": round(gbs, 1), "roofline_frac": round(gbs / 8000, 3)}
    eng.close()
    return out


def config3(n=8192, k=7, T=400):
    """5 classic layouts x 8192 kitchens, 15 pairing segments each, engines stepping concurrently on 5 streams."""
    threads = n // (2 * k + 1)
    segs = pairing_segments(threads, k)
    engines, acts, streams = [], [], []
    for name in CLASSIC_LAYOUTS:
        engines.append(OvercookedEngine(get_base_layout_params(name, 200), n, 0))
        acts.append(torch.randint(0, 6, (64, 2, n), device="cuda", dtype=torch.int32))
        streams.append(torch.cuda.Stream())

    def run():
        for eng, a, st in zip(engines, acts, streams):
            with torch.cuda.stream(st):
                eng.rollout(a, T)
    run()
    sec = timed(run)
    total_bytes = sum(algorithmic_bytes(e.size) * n for e in engines) * T
    out = {"what": "config 3: 5 classic layouts x 8192 kitchens, 15 pairing segments, 5 streams", "segments": len(segs),
           "kitchens_total": 5 * n, "us_per_joint_step": round(sec * 1e6 / T, 2),
           "agent_env_steps_per_s": round(2 * 5 * n * T / sec), "algorithmic_GBps": round(total_bytes / sec / 1e9, 1)}
    for e in engines:
        e.close()
    return out


def config3_group(n=8192, T=400, kpw=None, passes=0, wpc=0, layouts=None):
    """Config 3's env step as ONE launch per joint step (EngineGroup / oc_group_step): 5 classic layouts x 8192 kitchens."""
    from diverse_conventions_amd import EngineGroup
    layouts = list(layouts or CLASSIC_LAYOUTS)
    grp = EngineGroup.from_layouts(layouts, n, 0)
    if kpw or passes or wpc:
        grp.set_tuning(kpw, passes, wpc)
    acts = [torch.randint(0, 6, (64, 2, e.num_kitchens), device="cuda", dtype=torch.int32) for e in grp.engines]
    grp.rollout_timed(acts, 50)
    ms = min(grp.rollout_timed(acts, T) for _ in range(3))
    us = ms * 1e3 / T
    total_bytes = sum(algorithmic_bytes(e.size) * e.num_kitchens for e in grp.engines)
    kitchens = sum(e.num_kitchens for e in grp.engines)
    out = {"what": f"config 3 env step as ONE launch: {len(layouts)} layouts x {n} kitchens (oc_group_step)", "kitchens_total": kitchens,
           "us_per_joint_step": round(us, 2), "agent_env_steps_per_s": round(2 * kitchens / us * 1e6),
           "algorithmic_GBps": round(total_bytes / us / 1e3, 1), "roofline_frac": round(total_bytes / us / 1e3 / 8000, 3),
           "bytes_per_launch": total_bytes, "build": grp.build_info}
    grp.close(close_engines=True)
    return out


def config3_policies(n=8192, k=7, T=200, critics=True):
    """Config 3 with the policies in the loop: per layout 8 conventions' actors (+ one critic per segment and side) on
    the fused MFMA kernel + the env step, five layouts on five streams (the HBM-bound env kernel of one layout overlaps
    the MFMA-bound policy kernel of another)."""
    from diverse_conventions_amd import FusedPairedRollout, PackedNetworks
    threads = n // (2 * k + 1)
    G = 2 * k + 1
    gen = torch.Generator().manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=gen) * 0.1
    rollouts, streams = [], []
    for name in CLASSIC_LAYOUTS:
        params = get_base_layout_params(name, 200)
        W, H = params["width"], params["height"]
        P = (W - 2) * (H - 2)
        trunk = lambda: {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                         "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64)}
        actors = [dict(trunk(), **{"act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}) for _ in range(k + 1)]
        crit = [dict(trunk(), **{"v_out.weight": r(1, 64), "v_out.bias": r(1)}) for _ in range(G)] if critics else []
        nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=crit)
        eng = OvercookedEngine(params, threads * G, 0)
        rollouts.append(FusedPairedRollout(eng, nets, threads, k, critic_nets=[[k + 1 + g for g in range(G)]] * 2 if critics else None))
        streams.append(torch.cuda.Stream())

    def run():
        for ro, st in zip(rollouts, streams):
            with torch.cuda.stream(st):
                for _ in range(T):
                    ro.step()
    run()
    sec = timed(run)
    kitchens = sum(ro.engine.num_kitchens for ro in rollouts)
    out = {"what": "config 3 with policies in the loop: 5 classic layouts x 8190 kitchens, 8 conventions in 15 pairing segments, "
                   "fused MFMA policy kernel (actors" + (" + critics" if critics else "") + ") + env step, 5 streams",
           "kitchens_total": kitchens, "us_per_joint_step": round(sec * 1e6 / T, 1), "agent_env_steps_per_s": round(2 * kitchens * T / sec)}
    for ro in rollouts:
        ro.engine.close()
    return out


def config3_group_policies(n=8192, k=7, T=200, critics=True, partition="one_launch"):
    """Config 3 / 5 shape with the policies in the loop as 5 + 1 launches per joint step: every layout's fused policy launch side
    by side on CU-masked streams + ONE env launch for the five engines (GroupPairedRollout / oc_group_policy_rollout)."""
    from diverse_conventions_amd import EngineGroup, GroupPairedRollout, PackedNetworks
    threads = n // (2 * k + 1)
    G = 2 * k + 1
    gen = torch.Generator().manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=gen) * 0.1
    nets, engines = [], []
    for name in CLASSIC_LAYOUTS:
        params = get_base_layout_params(name, 200)
        W, H = params["width"], params["height"]
        P = (W - 2) * (H - 2)
        trunk = lambda: {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                         "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64)}
        actors = [dict(trunk(), **{"act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}) for _ in range(k + 1)]
        crit = [dict(trunk(), **{"v_out.weight": r(1, 64), "v_out.bias": r(1)}) for _ in range(G)] if critics else []
        nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=crit))
        engines.append(OvercookedEngine(params, threads * G, 0))
    grp = EngineGroup(engines)
    cn = [[[k + 1 + g for g in range(G)]] * 2] * 5 if critics else None
    if partition in ("streams", "cus"):                     # the per-engine-stream experiments (include/overcooked_experiments.h)
        from exp_group_streams import StreamedGroupPairedRollout
        ro = StreamedGroupPairedRollout(grp, nets, threads, k, mode=partition, critic_nets=cn)
    else:
        ro = GroupPairedRollout(grp, nets, threads, k, critic_nets=cn, partition=partition)
    ro.step(20)
    sec = timed(lambda: ro.step(T))
    kitchens = sum(e.num_kitchens for e in engines)
    out = {"what": f"config 3 shape with policies in the loop, {k + 1} conventions in {G} segments x 5 layouts: "
                   + {"cus": "5 policy launches + env launches, every engine on its own CU-masked stream",
                      "streams": "5 policy launches + env launches, every engine on its own stream",
                      "one_launch": "ONE policy launch over the five grids + ONE env launch",
                      False: "5 policy launches one after the other on one stream + ONE env launch"}[partition] + " per joint step"
                   + (" (actors + critics)" if critics else " (actors)"),
           "kitchens_total": kitchens, "us_per_joint_step": round(sec * 1e6 / T, 1), "agent_env_steps_per_s": round(2 * kitchens * T / sec),
           "cu_shares": getattr(ro, "cu_shares", None)}
    ro.close()
    grp.close(close_engines=True)
    return out


if __name__ == "__main__":
    if "--group-policies" in sys.argv:
        print(json.dumps(config3_policies(critics=False)), flush=True)
        for mode in (False, "streams", "cus", "one_launch"):
            print(json.dumps(config3_group_policies(critics=False, partition=mode)), flush=True)
        print(json.dumps(config3_policies(critics=True)), flush=True)
        for mode in (False, "streams", "one_launch"):
            print(json.dumps(config3_group_policies(critics=True, partition=mode)), flush=True)
        for mode in ("streams", False, "one_launch"):                                                       # config 5's per-GPU shape
            print(json.dumps(config3_group_policies(k=31, critics=True, partition=mode)), flush=True)
        sys.exit(0)
    if "--group" in sys.argv:
        print(json.dumps(config3()), flush=True)
        print(json.dumps(config3_group()), flush=True)
        print(json.dumps(config3_group(passes=2)), flush=True)
        print(json.dumps(config3_group(wpc=12)), flush=True)
        print(json.dumps(config3_group(wpc=12, passes=2)), flush=True)
        print(json.dumps(config3_group(wpc=10, passes=2)), flush=True)
        print(json.dumps(config3_group(kpw=[16, 8, 16, 16, 8], passes=2, wpc=12)), flush=True)
        print(json.dumps(config3_group(n=65536, T=100)), flush=True)
        print(json.dumps(config3_group(n=65536, T=100, passes=2, wpc=12)), flush=True)
        sys.exit(0)
    if "--policies" in sys.argv:
        print(json.dumps(config3_policies(critics=False)), flush=True)
        print(json.dumps(config3_policies(critics=True)), flush=True)
        sys.exit(0)
    for r in (single("cramped_room", 65536), time_major("cramped_room", 65536), single("cramped_room", 4096),
              single("asymmetric_advantages", 8192), single("asymmetric_advantages", 65536),
              single("coordination_ring", 65536), single("counter_circuit", 65536), config3()):
        print(json.dumps(r), flush=True)
