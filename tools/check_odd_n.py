#!/usr/bin/env python3
"""Parity of steps into time-major slots at kitchen counts whose second viewer plane starts off a 64-byte boundary (OC_HALO_R builds of
the module for fresh destinations, oc_step_body.h): every kitchen against the oracle, records included, in place in between.
    python tools/check_odd_n.py [layout] [kitchens ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from oracle import oracle
layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
counts = [int(v) for v in sys.argv[2:]] or [65535, 65534, 65533, 49153]
ok_all = True
for n in counts:
    params = get_base_layout_params(layout, 12)
    eng = OvercookedEngine(params, n, 0)
    ora = oracle.OracleVecEnv(params, n)
    W, H = params["width"], params["height"]
    T = 30
    slots = torch.zeros((T, 2, n, W, H, 20), dtype=torch.int8, device="cuda")
    rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    recs = torch.zeros((T, n, eng.export_states().shape[-1]), dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(n)
    ok = True
    for t in range(T):
        a = torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32, generator=g)
        if t % 5 == 4:
            obs, r, d = eng.step(a)                                 # in place now and then
        else:
            obs, r, d = eng.step(a, obs=slots[t], reward=rew[t], done=done[t], records=recs[t])
        oobs, orew, odone = ora.step(a.cpu().numpy())
        ok &= np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(r.cpu().numpy(), orew) and np.array_equal(d.cpu().numpy().astype(bool), np.asarray(odone).astype(bool))
        if t % 5 != 4:
            ok &= bool(torch.equal(recs[t], eng.export_states()))
        if not ok:
            print("MISMATCH at step", t); break
    print(f"{layout} {n} kitchens: {'bit-exact' if ok else 'WRONG'} over {T} steps ({eng.kernel_build_info})", flush=True)
    ok_all &= ok
    eng.close(); del slots
sys.exit(0 if ok_all else 1)
