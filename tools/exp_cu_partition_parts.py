#!/usr/bin/env python3
"""Where does the time go on CU-partitioned streams?  Five layouts (config 3 shape, actors only), shares as tools/exp_group_streams.py
picks them: each layout's policy launch and env launch timed alone on its masked stream, then all five side by side."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import CLASSIC_LAYOUTS, EngineGroup, OvercookedEngine, PackedNetworks, get_base_layout_params, native
from exp_group_streams import StreamedGroupPairedRollout

k, n = 7, 8192
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
    nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors))
    engines.append(OvercookedEngine(params, threads * G, 0))
grp = EngineGroup(engines)
ro = StreamedGroupPairedRollout(grp, nets, threads, k, mode="cus")
L = native.lib()
streams = [torch.cuda.ExternalStream(ro._streams[i]) for i in range(5)]
cus = ro.cu_shares


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)


def policy(i, masked=True):
    st = streams[i] if masked else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        sh = ro._shapes[i]
        native.check(L.oc_exp_policy_forward_cus(ctypes.byref(sh), ctypes.c_void_p(nets[i].image.data_ptr()), nets[i].n_nets,
                                             ctypes.c_void_p(engines[i].obs.data_ptr()), 2 * engines[i].num_kitchens, ctypes.c_void_p(ro.plans[i].tiles.data_ptr()),
                                             ro.plans[i].n_tiles, None, ctypes.c_void_p(ro.actions[i].data_ptr()), ctypes.c_void_p(ro.logp[i].data_ptr()), None, 1, 0, 0, None, None,
                                             cus[i] if masked else 0, ctypes.c_void_p(st.cuda_stream)))


def env(i, masked=True):
    L.oc_exp_set_cu_budget(engines[i]._handle, cus[i] if masked else 0)
    st = streams[i] if masked else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        engines[i].step(ro.actions[i])
    L.oc_exp_set_cu_budget(engines[i]._handle, 0)


out = {"cu_shares": cus, "tiles": [p.n_tiles for p in ro.plans]}
for i, name in enumerate(CLASSIC_LAYOUTS):
    out[name] = {"policy alone on its share": timed(lambda: policy(i)), "env alone on its share": timed(lambda: env(i)),
                 "policy alone, whole device": timed(lambda: policy(i, False)), "env alone, whole device": timed(lambda: env(i, False))}
out["all five policies side by side"] = timed(lambda: [policy(i) for i in range(5)])
out["all five env steps side by side"] = timed(lambda: [env(i) for i in range(5)])
out["all five (policy, env) side by side"] = timed(lambda: [(policy(i), env(i)) for i in range(5)])
out["all five policies, one stream, whole device"] = timed(lambda: [policy(i, False) for i in range(5)])
print(json.dumps(out, indent=1))
