#!/usr/bin/env python3
"""Where the microseconds of a Python-driven OvercookedMadrona.step() go (host side; the GPU is not the limit at 8192 kitchens):
each piece of VectorMultiAgentEnv.step timed on its own over 2000 iterations, queue drained between pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import OvercookedMadrona, RandomVectorAgent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = OvercookedMadrona("cramped_room", n, 0, debug_compile=False)
dev = env.sim.device
ego = torch.randint(0, 6, (64, n, 1), device=dev, dtype=torch.int64)
par = torch.randint(0, 6, (64, n, 1), device=dev, dtype=torch.int64)
tick = [0]
env.add_partner_agent(RandomVectorAgent(lambda: par[tick[0] % 64]))
env.reset()
K = 2000


def timed(name, fn):
    for t in range(50):
        fn(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K):
        fn(t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:58s} host {1e6 * (t1 - t0) / K:6.2f} us   with drain {1e6 * (t2 - t0) / K:6.2f} us", flush=True)


def full(t):
    tick[0] = t
    env.step(ego[t % 64])


acts = torch.stack([ego[0], par[0]])
timed("env.step(ego[t % 64])  (everything)", full)
timed("ego[t % 64]  (the caller's indexing)", lambda t: ego[t % 64])
timed("partner.get_action  (lambda: par[tick % 64])", lambda t: env.partners[0][0].get_action(None))
timed("torch.stack([ego, partner], out=actions)", lambda t: torch.stack([ego[0], par[0]], out=acts))
timed("env.sim.step(actions)  (checks + one ctypes call)", lambda t: env.sim.step(acts))
timed("env.n_step(actions)  (+ cached views, tuple)", lambda t: env.n_step(acts))
timed("env._update_players(rews, done)", lambda t: env._update_players(env.static_rewards, env.static_dones))
L, h = env.sim._L, env.sim._handle
import ctypes
from diverse_conventions_amd import native
pa, po, pr, pd = (ctypes.c_void_p(x.data_ptr()) for x in (acts, env.sim.obs, env.sim.reward, env.sim.done))
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
timed("bare L.oc_step(...) with prebuilt ctypes arguments", lambda t: L.oc_step(h, pa, native.OC_ACT_I64, po, pr, pd, None, st))
timed("torch.cuda.current_stream(dev).cuda_stream", lambda t: torch.cuda.current_stream(dev).cuda_stream)
