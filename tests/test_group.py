"""Several engines stepped by ONE launch (oc_group_*): the reference's five-layout set (train/env_utils.py:26 one simulator
per layout, train/test_all.sh:8; BASELINE configs 3 and 5).  CPU: the group module compiles for gfx950 and the launch
geometry balances the wavefronts; GPU: bit-exact against the oracle, against the engines stepped one by one, and for
every way the group can be configured."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import oracle

CLASSIC = ["simple", "unident_s", "random1", "random0", "random3"]


# ------------------------------------------------------------------------------------------------------------ CPU
def test_group_module_compiles_and_plan_gives_one_group_per_wavefront(layout_params):
    from diverse_conventions_amd import native
    ps = [layout_params[n] for n in CLASSIC]
    size, plan = native.jit_compile_check_group(ps, [8192] * 5)
    assert size > 50_000
    tiles = []
    for p, (kpw, blocks) in zip(ps, plan):
        S = p["width"] * p["height"]
        assert kpw % 4 == 0 and 4 <= kpw <= 64 and (kpw * S * 5) % 4 == 0
        assert blocks == -(-8192 // kpw), "config 3 fits one group per wavefront: nobody gets two"
        if S <= 32:
            assert kpw in (16, 32), "small grids take 16 or 32 kitchens per group (batched flush), whatever the tile balance says"
        else:
            tiles.append(kpw * S)
    assert sum(b for _, b in plan) <= 12 * 256
    even = sum(8192 * p["width"] * p["height"] for p in ps) / (12 * 256)         # tiles per wavefront of a perfect split
    # groups come in steps of 4 kitchens, so the best reachable budget sits within one step (4 x the largest grid) of an even split
    step = 4 * max(p["width"] * p["height"] for p in ps)
    assert max(tiles) <= even + step, f"unbalanced wavefronts: {tiles} tiles per group, {even:.0f} if split evenly"


def test_group_plan_large_batches_share_wavefronts_by_tiles(layout_params):
    from diverse_conventions_amd import native
    ps = [layout_params[n] for n in CLASSIC]
    _, plan = native.jit_compile_check_group(ps, [65536] * 5)
    blocks = np.array([b for _, b in plan], dtype=np.float64)
    tiles = np.array([65536 * p["width"] * p["height"] for p in ps], dtype=np.float64)
    assert blocks.sum() <= 12 * 256 and blocks.sum() >= 12 * 256 - 5
    assert np.abs(blocks / blocks.sum() - tiles / tiles.sum()).max() < 0.01


def test_group_module_dedupes_equal_layouts_and_takes_tiny_engines(layout_params):
    from diverse_conventions_amd import native
    p = layout_params["simple"]
    one, _ = native.jit_compile_check_group([p], [64])
    two, plan = native.jit_compile_check_group([p, p], [64, 64])
    assert plan[0] == plan[1]
    assert two < 1.2 * one, "two engines of one layout and tuning share one kernel body"
    _, plan = native.jit_compile_check_group([p, layout_params["unident_s"]], [1, 3])
    assert all(b >= 1 for _, b in plan)


def test_group_rejects_bad_arguments():
    import ctypes
    from diverse_conventions_amd import native
    L = native.lib()
    out = ctypes.c_void_p()
    assert L.oc_group_create(None, 1, ctypes.byref(out)) == -1
    arr = (ctypes.c_void_p * 9)()
    assert L.oc_group_create(arr, 9, ctypes.byref(out)) == -1 and b"1..8" in L.oc_last_error()
    assert L.oc_group_create(arr, 1, ctypes.byref(out)) == -1 and b"NULL" in L.oc_last_error()
    assert L.oc_group_step(None, None, 0, None, None, None, None, None) == -1
    assert L.oc_group_size(None) == 0 and L.oc_group_destroy(None) == 0


# ------------------------------------------------------------------------------------------------------------ GPU
def _mixed_actions(layout, T, n, seed, p_guided=0.85):
    g = load_golden(os.path.join(GOLDEN, f"guided_{layout}.npz"))["actions"]
    rng = np.random.default_rng(seed)
    base = np.broadcast_to(g[:T, :, None], (T, 2, n))
    rnd = rng.integers(0, 6, size=(T, 2, n))
    keep = rng.random((T, 1, n)) < p_guided
    return np.where(keep, base, rnd).astype(np.int32)


def _make(layout_params, names, counts, horizons=None):
    from diverse_conventions_amd import EngineGroup, OvercookedEngine
    horizons = horizons or [200] * len(names)
    params = [dict(layout_params[nm], horizon=h) for nm, h in zip(names, horizons)]
    engines = [OvercookedEngine(p, n, 0) for p, n in zip(params, counts)]
    return params, engines, EngineGroup(engines)


@pytest.mark.gpu
def test_group_step_is_one_launch_bit_exact_vs_oracle(layout_params):
    """The five classic layouts with ragged, different kitchen counts and different horizons in one launch per step:
    obs, reward, done of every kitchen at every step and the full state at the end, against the oracle."""
    counts, horizons = [1003, 517, 777, 64, 1290], [200, 150, 200, 90, 200]
    params, engines, grp = _make(layout_params, CLASSIC, counts, horizons)
    assert "one launch" in grp.build_info and "5 specialised bodies" in grp.build_info, grp.build_info
    T = 230
    acts = [_mixed_actions(nm, T, n, seed=3 + i) for i, (nm, n) in enumerate(zip(CLASSIC, counts))]
    oras = [oracle.OracleVecEnv(p, n, nthreads=8) for p, n in zip(params, counts)]
    for o, want in zip(grp.observe(), oras):
        assert np.array_equal(o.cpu().numpy(), want.encode())
    dacts = [torch.from_numpy(a).cuda() for a in acts]
    total = 0
    for t in range(T):
        obs, rew, done = grp.step([a[t] for a in dacts])
        for i, ora in enumerate(oras):
            oobs, orew, odone = ora.step(acts[i][t])
            assert np.array_equal(done[i].cpu().numpy(), odone), f"step {t} engine {i}: done"
            assert np.array_equal(rew[i].cpu().numpy(), orew), f"step {t} engine {i}: reward"
            assert np.array_equal(obs[i].cpu().numpy(), oobs), f"step {t} engine {i}: obs"
            total += int(orew[0].sum())
    assert total > 0
    for e, ora in zip(engines, oras):
        for x, y in zip(e.get_state(), ora.get_state()):
            assert np.array_equal(x, y)
    grp.close(close_engines=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "int64", "tuned", "nine_waves", "no_jit", "time_major", "misaligned", "dword_stores"])
def test_group_equals_engines_stepped_one_by_one(mode, layout_params, monkeypatch):
    """Whatever the configuration -- action dtype, launch geometry, the per-engine fallback, fresh destination slots
    (non-temporal stores), obs buffers that are only 4-byte aligned, the dword-store diagnostics -- a group step leaves
    exactly what stepping its engines one after the other leaves."""
    from diverse_conventions_amd import EngineGroup, OvercookedEngine
    names, counts, T = ["random3", "simple", "simple", "random0"], [300, 257, 64, 411], 40
    if mode == "no_jit":
        monkeypatch.setenv("OC_NO_JIT", "1")
    if mode == "dword_stores":
        monkeypatch.setenv("OC_DWORD_STORES", "1")
    params = [layout_params[n] for n in names]
    solo = [OvercookedEngine(p, n, 0) for p, n in zip(params, counts)]
    engines = [OvercookedEngine(p, n, 0) for p, n in zip(params, counts)]
    grp = EngineGroup(engines)
    if mode == "no_jit":
        assert "OC_NO_JIT" in grp.build_info
    else:
        assert "3 specialised bodies" in grp.build_info, grp.build_info       # the two `simple` engines share one
    if mode == "tuned":
        grp.set_tuning([4, 8, 52, 12], passes_per_flush=2)
    if mode == "nine_waves":
        grp.set_tuning(None, 0, 9)
    rng = np.random.default_rng(8)
    dt = torch.int64 if mode == "int64" else torch.int32
    acts = [torch.from_numpy(rng.integers(0, 6, size=(T, 2, n))).to(dt).cuda() for n in counts]
    if mode == "time_major":
        slots = [torch.zeros((T,) + tuple(e.obs.shape), dtype=torch.int8, device="cuda") for e in engines]
    if mode == "misaligned":
        raw = [torch.zeros(e.obs.numel() + 16, dtype=torch.int8, device="cuda") for e in engines]
        odd = [r[4:4 + e.obs.numel()].view(e.obs.shape) for r, e in zip(raw, engines)]
        assert all(o.data_ptr() % 16 == 4 for o in odd)
    for t in range(T):
        kw = {}
        if mode == "time_major":
            kw["obs"] = [s[t] for s in slots]
        if mode == "misaligned":
            kw["obs"] = odd
        obs, rew, done = grp.step([a[t] for a in acts], **kw)
        for i, s in enumerate(solo):
            so, sr, sd = s.step(acts[i][t])
            assert torch.equal(obs[i], so) and torch.equal(rew[i], sr) and torch.equal(done[i], sd), f"step {t} engine {i}"
    for e, s in zip(engines, solo):
        assert torch.equal(e.export_states(), s.export_states())
    grp.close(close_engines=True)
    for s in solo:
        s.close()


@pytest.mark.gpu
def test_config3_shape_in_one_launch(layout_params):
    """BASELINE config 3 as one launch per joint step: the five classic layouts x 8192 kitchens, 8 conventions in the
    reference's 15 pairing segments (every (policy, side) pair its own action stream); 256 sampled kitchens per layout
    against the oracle at every step, all kitchens' rewards/dones against a second engine set stepped one by one."""
    from diverse_conventions_amd import pairing_segments
    n, k, T = 8192, 7, 210
    threads = n // (2 * k + 1)
    segs = pairing_segments(threads, k)
    rng = np.random.default_rng(9)
    tapes = rng.integers(0, 6, size=(k + 1, 2, T, threads)).astype(np.int32)
    acts = np.full((T, 2, n), 4, dtype=np.int32)
    for s in segs:
        acts[:, 0, s.start:s.stop] = tapes[s.ego_policy, 0]
        acts[:, 1, s.start:s.stop] = tapes[s.partner_policy, 1]
    sample = np.sort(rng.choice(n, size=256, replace=False))
    params, engines, grp = _make(layout_params, CLASSIC, [n] * 5)
    assert "one launch" in grp.build_info, grp.build_info
    oras = [oracle.OracleVecEnv(p, len(sample), nthreads=8) for p in params]
    dacts = torch.from_numpy(acts).cuda()
    idx = torch.from_numpy(sample).cuda()
    for t in range(T):
        obs, rew, done = grp.step([dacts[t]] * 5)
        for i, ora in enumerate(oras):
            oobs, orew, odone = ora.step(acts[t][:, sample])
            assert np.array_equal(obs[i][:, idx].cpu().numpy(), oobs), f"step {t} layout {CLASSIC[i]}"
            assert np.array_equal(rew[i][:, idx].cpu().numpy(), orew) and np.array_equal(done[i][idx].cpu().numpy(), odone)
        if t in (199, 200):
            assert all(int(d.sum()) == (n if t == 199 else 0) for d in done)
    grp.close(close_engines=True)


@pytest.mark.gpu
def test_encode_trajectory_is_the_per_step_encode(layout_params):
    """oc_encode_trajectory: T strided record blocks -> T time slots of a larger tensor in one launch, equal to one
    oc_encode_states_strided call per time step; neighbouring kitchens of the destination stay untouched."""
    from diverse_conventions_amd import OvercookedEngine
    for name, n, T in (("unident_s", 37, 9), ("simple", 200, 5), ("random1", 5, 33)):
        eng = OvercookedEngine(layout_params[name], n, 0)
        r4 = eng.record_bytes // 4
        rng = np.random.default_rng(n)
        acts = torch.from_numpy(rng.integers(0, 6, size=(T, 2, n)).astype(np.int32)).cuda()
        padded = torch.zeros((T, n * r4 + 8), dtype=torch.int32, device="cuda")            # rows with a gap: a strided source
        recs = padded[:, :n * r4].view(T, n, r4)
        for t in range(T):
            eng.step(acts[t])
            recs[t].copy_(eng.export_states())
        n_total, off = n + 11, 7
        want = torch.full((T, 2, n_total, eng.width, eng.height, 20), -7, dtype=torch.int8, device="cuda")
        got = want.clone()
        for t in range(T):
            eng.encode_states_into(recs[t].contiguous(), want[t], off)
        eng.encode_trajectory_into(recs, got, off)
        assert torch.equal(got, want), name
        assert int((got[:, :, :off] != -7).sum()) == 0 and int((got[:, :, off + n:] != -7).sum()) == 0
        eng.close()


@pytest.mark.gpu
def test_the_group_binding_shown_in_integration_md_runs(layout_params):
    """INTEGRATION.md 2b stepped against the simulator stub of section 2, both executed as written: a SimulatorGroup step
    over three layouts leaves in every simulator's persistent tensors what its own step() leaves."""
    import re
    from conftest import ROOT
    from diverse_conventions_amd import native
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    ns = {}
    for name in ("envs/overcooked_hip_binding.py", "envs/overcooked_hip_group.py"):          # two files -> two module namespaces
        block = re.search(r"```python\n# " + re.escape(name) + r"\n(.*?)```", text, re.S).group(1)
        mod = {}
        exec(compile(block.replace('"diverse-conventions_amd/csrc/liboc_hip.so"', repr(native.LIB_PATH)), "INTEGRATION.md:" + name, "exec"), mod)
        ns.update({k: v for k, v in mod.items() if k in ("SimplecookedSimulator", "SimulatorGroup")})
    keys = ("terrain", "height", "width", "num_players", "start_player_x", "start_player_y", "placement_in_pot_rew", "dish_pickup_rew",
            "soup_pickup_rew", "recipe_values", "recipe_times", "horizon")
    make = lambda nm, n: ns["SimplecookedSimulator"](exec_mode=None, gpu_id=0, num_worlds=n, debug_compile=False, **{k: layout_params[nm][k] for k in keys})
    names, counts = ["simple", "random3", "unident_s"], [130, 77, 260]
    solo = [make(nm, n) for nm, n in zip(names, counts)]
    sims = [make(nm, n) for nm, n in zip(names, counts)]
    grp = ns["SimulatorGroup"](sims)
    rng = np.random.default_rng(0)
    for t in range(30):
        for a, b, n in zip(solo, sims, counts):
            act = torch.from_numpy(rng.integers(0, 6, size=(2, n, 1)).astype(np.int32)).cuda()
            a.actions.copy_(act); b.actions.copy_(act)
            a.step()
        grp.step()
        for a, b in zip(solo, sims):
            assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), t


@pytest.mark.gpu
@pytest.mark.parametrize("partition", [False, "one_launch"])
@pytest.mark.parametrize("deterministic", [True, False])
def test_group_rollout_with_policies_equals_per_engine_rollouts(partition, deterministic, layout_params, monkeypatch):
    """Policies in the loop over three layouts: per joint step three policy launches on the launch stream, or ONE policy launch
    over the three grids ("one_launch"), and ONE env launch.  Every engine's collected trajectory -- actions, log-probs, values,
    observations, rewards, dones -- must be exactly what its own FusedPairedRollout collects with the same seed, and the env
    oracle driven by the recorded actions must reproduce the observations.  (Cooperative blocks off: two plans over the same rows
    share different blocks, and a shared block sums its fc1 partials in another order -- tests/test_fused_policy.py.)"""
    monkeypatch.setenv("OC_POLICY_COOP", "0")
    from diverse_conventions_amd import (EngineGroup, FusedPairedRollout, GroupPairedRollout, OvercookedEngine, PackedNetworks,
                                         TrajectoryBuffer)
    names, threads, k, T = ["simple", "random0", "unident_s"], [48, 33, 20], 2, 14
    G = 2 * k + 1
    rng = np.random.default_rng(5)
    r = lambda *s: (rng.standard_normal(s) * 0.1).astype(np.float32)
    nets, critic_of = [], []
    for nm in names:
        p = layout_params[nm]
        W, H = p["width"], p["height"]
        P = (W - 2) * (H - 2)
        trunk = lambda: {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                         "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64)}
        actors = [dict(trunk(), **{"act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}) for _ in range(k + 1)]
        critics = [dict(trunk(), **{"v_out.weight": r(1, 64), "v_out.bias": r(1)}) for _ in range(2)]
        nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=critics))
        critic_of.append([[k + 1 + (g % 2) for g in range(G)], [k + 1 + ((g + 1) % 2) for g in range(G)]])
    make = lambda: [OvercookedEngine(layout_params[nm], th * G, 0) for nm, th in zip(names, threads)]
    seed = 77
    # reference: every engine on its own, one after the other
    want = []
    for i, e in enumerate(make()):
        ro = FusedPairedRollout(e, nets[i], threads[i], k, critic_nets=critic_of[i], deterministic=deterministic,
                                seed=(seed + i * 0x9E3779B9) & 0xFFFFFFFF)
        tr = TrajectoryBuffer(e, T)
        lp, va = torch.zeros((T, 2, e.num_kitchens), device="cuda"), torch.zeros((T, 2, e.num_kitchens), device="cuda")
        ro.collect(tr, logp=lp, values=va)
        want.append((tr, lp, va))
    engines = make()
    grp = EngineGroup(engines)
    gro = GroupPairedRollout(grp, nets, threads, k, critic_nets=critic_of, deterministic=deterministic, seed=seed, partition=partition)
    trajs = [TrajectoryBuffer(e, T, store_states=(i != 1)) for i, e in enumerate(engines)]     # (a group may record some engines only)
    lps = [torch.zeros((T, 2, e.num_kitchens), device="cuda") for e in engines]
    vas = [torch.zeros((T, 2, e.num_kitchens), device="cuda") for e in engines]
    gro.collect(trajs, logps=lps, values=vas)
    torch.cuda.synchronize()
    for i, (tr, lp, va) in enumerate(want):
        assert torch.equal(trajs[i].actions, tr.actions), f"engine {i}: actions"
        assert torch.equal(trajs[i].obs, tr.obs) and torch.equal(trajs[i].reward, tr.reward) and torch.equal(trajs[i].done, tr.done)
        assert torch.equal(lps[i], lp) and torch.equal(vas[i], va)
        ora = oracle.OracleVecEnv(layout_params[names[i]], engines[i].num_kitchens)
        acts = tr.actions.cpu().numpy()
        for t in range(T):
            oobs, orew, odone = ora.step(acts[t])
            assert np.array_equal(trajs[i].obs[t + 1].cpu().numpy(), oobs) and np.array_equal(trajs[i].reward[t].cpu().numpy(), orew)
    if not deterministic:
        assert len(np.unique(want[0][0].actions.cpu().numpy())) == 6
    # compact records written by the joint env launch itself (oc_group_step's d_records): every slot re-featurises to its observation
    from diverse_conventions_amd import TrajectoryExchange
    for i in (0, 2):
        assert torch.equal(TrajectoryExchange().featurize_all(engines[i], trajs[i].records.unsqueeze(0)), trajs[i].obs), i
        assert torch.equal(trajs[i].records[T], engines[i].export_states())
    # in-place stepping continues the draw counter where the collection stopped
    obs, rew, done = gro.step(3)
    assert gro.steps_done == T + 3 and obs[0].data_ptr() == engines[0].obs.data_ptr()
    gro.close()
    grp.close(close_engines=True)
    for tr, _, _ in want:
        tr.engine.close()


@pytest.mark.gpu
def test_group_with_other_player_counts_is_refused_by_the_policy_rollouts(layout_params):
    """A group may hold engines with 1, 3 or 4 players (its joint step then runs them on the plain kernels), but the fused policy
    kernel reads 2-player observations and the paired rollouts' action / reward buffers are [2, N]: both C entry points and the
    Python class refuse such a group instead of reading 30-channel planes as 20-channel rows."""
    import ctypes
    from diverse_conventions_amd import EngineGroup, GroupPairedRollout, OvercookedEngine, PackedNetworks, get_base_layout_params, native
    from diverse_conventions_amd.fused_policy import _shape
    rng = np.random.default_rng(0)
    r = lambda *sh: (rng.standard_normal(sh) * 0.1).astype(np.float32)
    p4 = get_base_layout_params("multiplayer_schelling", 200)
    two = OvercookedEngine(layout_params["simple"], 96, 0)
    four = OvercookedEngine(p4, 96, 0)
    assert four.num_players == 4
    grp = EngineGroup([two, four])
    nets = []
    for W, H in ((5, 4), (p4["width"], p4["height"])):
        P = (W - 2) * (H - 2)
        sd = {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P), "base.cnn.cnn.3.bias": r(64),
              "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64), "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}
        nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[sd]))
    with pytest.raises(native.OvercookedNativeError, match="2-player"):
        GroupPairedRollout(grp, nets, [96, 96], 0)
    # the C entry points themselves, handed well-formed slots
    L = native.lib()
    shapes = [_shape(nt.W, nt.H, 6, nt.digits) for nt in nets]
    tiles = torch.tensor([[0, 192, 0, 1]], dtype=torch.int32, device="cuda")
    acts = [torch.zeros((2, 96), dtype=torch.int32, device="cuda") for _ in range(2)]
    slots = (native.OcGroupPolicySlot * 2)()
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    for i, e in enumerate((two, four)):
        sl = slots[i]
        sl.shape = ctypes.pointer(shapes[i])
        sl.d_packed, sl.n_nets, sl.d_tiles, sl.n_tiles = vp(nets[i].image), 1, vp(tiles), 1
        sl.d_obs, sl.d_actions, sl.d_reward, sl.d_done = vp(e.obs), vp(acts[i]), vp(e.reward), vp(e.done)
    before = four.export_states().clone()
    rc = L.oc_group_policy_rollout(grp._handle, slots, 2, 1, 0, 0, 0, None)
    assert rc == -2 and "2-player" in L.oc_last_error().decode()
    rc = L.oc_group_policy_rollout_multi(grp._handle, slots, vp(tiles), 1, 1, 4, 2, 1, 0, 0, 0, None)
    assert rc in (-1, -2) and "2-player" in L.oc_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(four.export_states(), before)              # nothing was stepped
    grp.close(close_engines=True)
