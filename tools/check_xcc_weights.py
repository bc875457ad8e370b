#!/usr/bin/env python3
"""Parity of the per-XCD group sizes (OC_XCC_WEIGHTS, diagnostics build) against the oracle: every kitchen of a ragged batch, in place and
into time-major slots, records included.   python tools/check_xcc_weights.py [layout] [kitchens] [kpw] [w0,...,w7]"""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65_531
kpw = int(sys.argv[3]) if len(sys.argv) > 3 else 24
os.environ["OC_XCC_WEIGHTS"] = sys.argv[4] if len(sys.argv) > 4 else "19,13,17,15,16,0,23,1"
from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
from oracle import oracle
params = get_base_layout_params(layout, 30)
eng = OvercookedEngine(params, n, 0)
eng.set_tuning(kpw, 4, 8)
print(eng.kernel_build_info)
ora = oracle.OracleVecEnv(params, n)
W, H = params["width"], params["height"]
T = 40
slots = torch.zeros((T, 2, n, W, H, 20), dtype=torch.int8, device="cuda")
rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
recs = torch.zeros((T, n, eng.export_states().shape[-1]), dtype=torch.int32, device="cuda")
ok = True
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(T):
    a = torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32, generator=g)
    if t % 2:
        obs, r, d = eng.step(a)                                     # in place
    else:
        obs, r, d = eng.step(a, obs=slots[t], reward=rew[t], done=done[t], records=recs[t])
    oobs, orew, odone = ora.step(a.cpu().numpy())
    ok &= np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(r.cpu().numpy(), orew) and np.array_equal(d.cpu().numpy().astype(bool), np.asarray(odone).astype(bool))
    if t % 2 == 0:
        ok &= bool(torch.equal(recs[t], eng.export_states()))
    if not ok:
        print("MISMATCH at step", t); break
print(f"{layout} {n} kitchens, kpw {kpw}, weights {os.environ['OC_XCC_WEIGHTS']}: {'bit-exact' if ok else 'WRONG'} over {T} steps")
sys.exit(0 if ok else 1)
