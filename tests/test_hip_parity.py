"""Parity tests proper: the HIP engine, called through the C ABI, against (a) the committed golden
fixtures generated from the reference Python and (b) the CPU oracle on seeded action streams.
Bar: bit-exact (integer / byte work).  All tests need a real MI355X."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files, golden_layout_of, load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu

TWO_PLAYER = lambda p: p["num_players"] == 2
FIXTURES = golden_files("random_") + golden_files("guided_") + golden_files("micro_") + golden_files("scripted_")


def _engine(params, n):
    from diverse_conventions_amd import OvercookedEngine
    return OvercookedEngine(params, n, 0)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_replays_reference_fixture(path, layout_params):
    """HIP engine vs outputs recorded from the reference itself: obs, reward, done, full state, every step."""
    rec = load_golden(path)
    params = layout_params[golden_layout_of(path, rec)]
    eng = _engine(params, 1)
    eng.set_state(rec["init_tiles"][None], rec["init_players"][None], rec["init_timestep"][None])
    assert np.array_equal(eng.observe().cpu().numpy()[:, 0], rec["init_obs"])
    T = len(rec["actions"])
    acts = torch.from_numpy(rec["actions"].astype(np.int32)).cuda().reshape(T, 2, 1)
    for t in range(T):
        obs, rew, done = eng.step(acts[t])
        assert np.array_equal(obs.cpu().numpy()[:, 0], rec["obs"][t]), f"step {t}: obs"
        assert np.all(rew.cpu().numpy()[:, 0] == rec["reward"][t]), f"step {t}: reward"
        assert int(done.cpu()[0]) == int(rec["done"][t]), f"step {t}: done"
        tiles, players, ts = eng.get_state()
        assert np.array_equal(tiles[0], rec["tiles"][t]), f"step {t}: tiles"
        assert np.array_equal(players[0], rec["players"][t]), f"step {t}: players"
        assert int(ts[0]) == int(rec["timestep"][t]), f"step {t}: timestep"
    eng.close()


FUZZ = golden_files("fuzz_")


@pytest.mark.parametrize("path", FUZZ, ids=[os.path.basename(p)[:-4] for p in FUZZ])
def test_hip_matches_reference_on_arbitrary_states(path, layout_params):
    """500 arbitrary (also unreachable) states recorded from the reference, loaded as 500 kitchens:
    cooking soups on counters, idle full soups, foreign objects in pots, both players on one tile ..."""
    z = load_golden(path)
    params = layout_params[os.path.basename(path)[5:-4]]
    n = len(z["tiles"])
    for kpw in (0, 4):
        eng = _engine(params, n)
        if kpw:
            eng.set_tuning(kpw, 1, 3)
        eng.set_state(z["tiles"], z["players"], z["timestep"])
        assert np.array_equal(eng.observe().cpu().numpy(), np.moveaxis(z["init_obs"], 0, 1))
        obs, rew, done = eng.step(torch.from_numpy(np.ascontiguousarray(z["actions"].T)).cuda())
        tiles, players, ts = eng.get_state()
        assert np.array_equal(done.cpu().numpy().astype(np.uint8), z["done"])
        assert np.array_equal(rew.cpu().numpy()[0], z["reward_per_player"].sum(axis=1))
        assert np.array_equal(tiles, z["n_tiles"]) and np.array_equal(players, z["n_players"])
        assert np.array_equal(ts, z["n_timestep"])
        assert np.array_equal(obs.cpu().numpy(), np.moveaxis(z["obs"], 0, 1))
        eng.close()


def _mixed_actions(layout, T, n, seed, p_guided=0.85):
    """Per-kitchen action streams: the fixture's guided stream (which cooks and delivers) with
    per-kitchen random substitutions, so kitchens diverge but keep doing useful work."""
    g = load_golden(os.path.join(GOLDEN, f"guided_{layout}.npz"))["actions"]
    rng = np.random.default_rng(seed)
    base = np.broadcast_to(g[:T, :, None], (T, 2, n))
    rnd = rng.integers(0, 6, size=(T, 2, n))
    keep = rng.random((T, 1, n)) < p_guided
    return np.where(keep, base, rnd).astype(np.int32)


@pytest.mark.parametrize("layout", ["simple", "unident_s", "random1", "random0", "random3", "mdp_test",
                                    "simple_tomato", "corridor", "scenario2_s", "six_pots", "tiny_mixed", "wide_hall"])
def test_hip_vs_oracle_batch(layout, layout_params):
    """4099 kitchens (ragged last wave), 450 steps crossing two horizon resets."""
    params = layout_params[layout]
    n, T = 4099, 450
    acts = _mixed_actions(layout, T, n, seed=11)
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    eng = _engine(params, n)
    assert np.array_equal(eng.observe().cpu().numpy(), ora.encode())
    dacts = torch.from_numpy(acts).cuda()
    total = 0
    for t in range(T):
        obs, rew, done = eng.step(dacts[t])
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(rew.cpu().numpy(), orew), f"step {t}: reward"
        assert np.array_equal(done.cpu().numpy(), odone), f"step {t}: done"
        assert np.array_equal(obs.cpu().numpy(), oobs), f"step {t}: obs"
        total += int(orew[0].sum())
        if t % 50 == 49 or t == T - 1:
            for x, y in zip(eng.get_state(), ora.get_state()):
                assert np.array_equal(x, y), f"step {t}: state"
    assert total > 0          # the streams really cook and deliver
    eng.close()


@pytest.mark.parametrize("kpw,passes,wpc", [(4, 1, 1), (8, 2, 4), (16, 4, 8), (32, 8, 2), (40, 3, 3), (12, 5, 16), (16, 2, 32)])
def test_tuning_does_not_change_results(kpw, passes, wpc, layout_params):
    params = layout_params["random1"]            # S = 25: odd record padding, non-16B kitchen planes
    n, T = 1003, 230
    acts = _mixed_actions("random1", T, n, seed=5)
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    eng = _engine(params, n)
    eng.set_tuning(kpw, passes, wpc)
    dacts = torch.from_numpy(acts).cuda()
    for t in range(T):
        obs, rew, done = eng.step(dacts[t])
        oobs, orew, odone = ora.step(acts[t])
        if t % 23 == 0 or t >= 198:
            assert np.array_equal(obs.cpu().numpy(), oobs), t
            assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), t
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()


def test_one_wave_per_kitchen_geometry(layout_params):
    """kitchens_per_wave = 1 (the north star's first sketch, kept as a measurable geometry) is as exact as any."""
    from diverse_conventions_amd.native import OvercookedNativeError
    params = layout_params["simple"]
    n, T = 301, 205
    acts = _mixed_actions("simple", T, n, seed=77)
    ora = oracle.OracleVecEnv(params, n)
    eng = _engine(params, n)
    eng.set_tuning(1, 1, 32)
    dacts = torch.from_numpy(acts).cuda()
    for t in range(T):
        obs, rew, done = eng.step(dacts[t])
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew), t
    eng.close()
    odd = _engine(layout_params["random1"], 8)            # S = 25: groups of 1 kitchen would start on 4-byte boundaries
    with pytest.raises(OvercookedNativeError):
        odd.set_tuning(1, 1, 8)
    odd.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 63, 64, 65])
def test_small_and_ragged_sizes(n, layout_params, monkeypatch):
    if n % 2:
        monkeypatch.setenv("OC_ALIGNED_PASSES", "1")      # (diagnostics build only: odd sizes then also exercise the kitchen-aligned pass mode)
    for layout in ("simple", "random0", "unident_s"):
        params = layout_params[layout]
        acts = _mixed_actions(layout, 210, n, seed=n)
        ora = oracle.OracleVecEnv(params, n)
        eng = _engine(params, n)
        dacts = torch.from_numpy(acts).cuda()
        for t in range(210):
            obs, rew, done = eng.step(dacts[t])
            oobs, orew, odone = ora.step(acts[t])
            assert np.array_equal(obs.cpu().numpy(), oobs), (layout, t)
            assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone)
        eng.close()


@pytest.mark.parametrize("stores", ["wide", "dword"])
@pytest.mark.parametrize("offset", [4, 8, 12])
def test_unaligned_destination_and_int64_actions(stores, offset, layout_params, monkeypatch):
    """obs written into a caller buffer that is only 4-byte aligned -- 16-byte stores on dword-aligned addresses, the
    default, and the dword-store kernels (OC_DWORD_STORES=1) -- with canary bytes on both sides; actions as int64
    [P, N, 1] exactly as torch.stack of sampled actions gives them."""
    if stores == "dword":
        monkeypatch.setenv("OC_DWORD_STORES", "1")
    params = layout_params["simple"]
    n, T = 515, 60
    acts = _mixed_actions("simple", T, n, seed=2)
    ora = oracle.OracleVecEnv(params, n)
    eng = _engine(params, n)
    raw = torch.zeros(eng.obs.numel() + 64, dtype=torch.int8, device="cuda")
    view = raw[offset:offset + eng.obs.numel()].view(eng.obs.shape)
    assert view.data_ptr() % 16 == offset
    for t in range(T):
        a64 = torch.from_numpy(acts[t]).cuda().to(torch.int64).unsqueeze(-1)
        obs, rew, done = eng.step(a64, obs=view)
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), oobs), t
        assert np.array_equal(rew.cpu().numpy(), orew)
    assert int(raw[:offset].abs().sum()) == 0 and int(raw[offset + eng.obs.numel():].abs().sum()) == 0   # no stray writes
    eng.close()


def test_out_of_range_action_is_stay(layout_params):
    params = layout_params["simple"]
    eng = _engine(params, 4)
    ora = oracle.OracleVecEnv(params, 4)
    bad = torch.tensor([[7, -1, 6, 100], [2, 3, -5, 9]], dtype=torch.int32, device="cuda")
    ok = np.array([[4, 4, 4, 4], [2, 3, 4, 4]], dtype=np.int32)
    obs, rew, done = eng.step(bad)
    oobs, _, _ = ora.step(ok)
    assert np.array_equal(obs.cpu().numpy(), oobs)
    eng.close()


def test_state_roundtrip_encode_and_exchange(layout_params):
    """get/set round trip, device export/import, and featurising exported records elsewhere
    (the receive side of a compact trajectory exchange) all agree with the oracle."""
    params = layout_params["unident_s"]
    n = 777
    acts = _mixed_actions("unident_s", 120, n, seed=9)
    ora = oracle.OracleVecEnv(params, n, nthreads=4)
    eng = _engine(params, n)
    dacts = torch.from_numpy(acts).cuda()
    for t in range(120):
        eng.step(dacts[t]); ora.step(acts[t])
    recs = eng.export_states()
    other = _engine(params, 10)
    enc = other.encode_states(recs)                       # 777 records through a 10-kitchen engine
    assert np.array_equal(enc.cpu().numpy(), ora.encode())
    tiles, players, ts = eng.get_state()
    eng2 = _engine(params, n)
    eng2.set_state(tiles, players, ts)
    assert np.array_equal(eng2.observe().cpu().numpy(), ora.encode())
    eng3 = _engine(params, n)
    eng3.import_states(recs)
    for e in (eng, eng2, eng3):
        obs, rew, done = e.step(dacts[0])
    oobs, orew, _ = ora.step(acts[0])
    for e in (eng, eng2, eng3):
        assert np.array_equal(e.obs.cpu().numpy(), oobs) and np.array_equal(e.reward.cpu().numpy(), orew)
        e.close()
    other.close()


def test_reset_all_and_rollout_time_major(layout_params):
    params = layout_params["random3"]
    n, T = 300, 205
    acts = _mixed_actions("random3", T, n, seed=4)
    ora = oracle.OracleVecEnv(params, n)
    eng = _engine(params, n)
    dacts = torch.from_numpy(acts).cuda()
    eng.step(dacts[0]); eng.step(dacts[1])
    start = eng.reset_all().cpu().numpy()
    assert np.array_equal(start, ora.reset_all())
    obs = torch.zeros((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
    rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    eng.rollout(dacts, T, obs=obs, reward=rew, done=done)     # T launches from C, time-major outputs
    for t in range(T):
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs[t].cpu().numpy(), oobs), t
        assert np.array_equal(rew[t].cpu().numpy(), orew) and np.array_equal(done[t].cpu().numpy(), odone)
    assert int(done[199].sum()) == n and int(done.sum()) == n
    eng.close()


def test_time_major_steps_of_a_large_small_grid_batch_use_their_own_geometry(layout_params):
    """49 152 cramped_room kitchens take 32 kitchens per group in place; steps into fresh destinations run with 16 per group on
    eight wavefronts per CU from a module of their own, built on the first such step: in-place steps, time-major steps and in-place
    steps again against the oracle on a sample of kitchens from every part of the batch (kitchens are independent, so the oracle
    steps just those)."""
    params = layout_params["simple"]
    n, T = 49152, 9
    rng = np.random.default_rng(12)
    pick = np.unique(np.concatenate([np.arange(70), rng.integers(0, n, 300), np.arange(n - 70, n), np.arange(24544, 24608)]))
    acts = rng.integers(0, 6, size=(3 * T, 2, n)).astype(np.int32)
    dacts = torch.from_numpy(acts).cuda()
    eng = _engine(params, n)
    if "specialised" in eng.kernel_build_info:       # (OC_NO_JIT=1 runs this file on the generic ahead-of-time kernel: no per-engine geometry string)
        assert "kpw 32" in eng.kernel_build_info
    eng.prepare_kernels()                        # the time-major and fused-rollout modules, ahead of their first use (idempotent)
    eng.prepare_kernels()
    ora = oracle.OracleVecEnv(params, len(pick))
    slots = torch.zeros((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
    rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    t = 0
    for phase in ("in place", "time major", "in place"):
        if phase == "time major":
            eng.rollout(dacts[t:t + T], T, obs=slots, reward=rew, done=done)
        for i in range(T):
            oobs, orew, odone = ora.step(acts[t + i][:, pick])
            if phase == "time major":
                got, grew, gdone = slots[i], rew[i], done[i]
            else:
                got, grew, gdone = eng.step(dacts[t + i])
            assert np.array_equal(got[:, pick].cpu().numpy(), oobs), (phase, i)
            assert np.array_equal(grew[:, pick].cpu().numpy(), orew) and np.array_equal(gdone[pick].cpu().numpy(), odone)
        t += T
    eng.close()


@pytest.mark.parametrize("layout", ["unident_s", "random1"])
def test_benched_shapes_sampled_parity_in_place_and_time_major(layout, layout_params):
    """The shapes bench.py times beyond the headline, at their full size: asymmetric_advantages x 65 536 (`config4_split` at one GPU)
    and coordination_ring x 65 536, stepped in place, into time-major slots (the fresh-destination module with its own geometry
    where there is one) and in place again.  Launch geometry is chosen from the kitchen count, so parity at 8184 kitchens says
    nothing about this size: (1) 512 sampled kitchens with their own guided streams against the oracle at every step;
    (2) replication: every kitchen outside the sample gets kitchen 0's stream and must stay identical to it."""
    params = layout_params[layout]
    n, T = 65536, 8
    rng = np.random.default_rng(31)
    sample = np.sort(rng.choice(np.arange(1, n), size=512, replace=False))
    own = _mixed_actions(layout, 3 * T, 512, seed=5)
    lead = _mixed_actions(layout, 3 * T, 1, seed=6)
    acts = np.repeat(lead, n, axis=2)
    acts[:, :, sample] = own
    dacts = torch.from_numpy(acts).cuda()
    others = np.setdiff1d(np.arange(n), sample)
    eng = _engine(params, n)
    ora = oracle.OracleVecEnv(params, 512)
    slots = torch.zeros((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
    rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    t = 0
    for phase in ("in place", "time major", "in place"):
        if phase == "time major":
            eng.rollout(dacts[t:t + T], T, obs=slots, reward=rew, done=done)
        for i in range(T):
            oobs, orew, odone = ora.step(own[t + i])
            got, grew, gdone = (slots[i], rew[i], done[i]) if phase == "time major" else eng.step(dacts[t + i])
            assert np.array_equal(got[:, sample].cpu().numpy(), oobs), (phase, i)
            assert np.array_equal(grew[:, sample].cpu().numpy(), orew) and np.array_equal(gdone[sample].cpu().numpy(), odone)
            rest = got[:, others]
            assert bool((rest == rest[:, :1]).all()), (phase, i, "replication")
            assert bool((grew[:, others] == grew[:, others[:1]]).all())
        t += T
    eng.close()


def test_full_size_properties(layout_params):
    """65 536 kitchens (BASELINE.json's size): size-independent checks.
    (1) replication: kitchens fed identical action streams stay identical to kitchen 0;
    (2) sampled parity: 512 scattered kitchens with their own streams match the oracle stepped on
        exactly those streams; (3) every kitchen reports done exactly at steps 200, 400."""
    params = layout_params["simple"]
    n, T = 65536, 402
    rng = np.random.default_rng(21)
    sample = np.sort(rng.choice(np.arange(1, n), size=512, replace=False))
    own = _mixed_actions("simple", T, 512, seed=8)
    base = _mixed_actions("simple", T, 1, seed=3)[:, :, 0]
    eng = _engine(params, n)
    ora = oracle.OracleVecEnv(params, 512, nthreads=8)
    ref0 = oracle.OracleVecEnv(params, 1)
    idx = torch.from_numpy(sample).cuda()
    mask = torch.ones(n, dtype=torch.bool, device="cuda"); mask[idx] = False
    for t in range(T):
        a = torch.from_numpy(np.repeat(base[t][:, None], n, axis=1)).cuda().to(torch.int32)
        a[:, idx] = torch.from_numpy(own[t]).cuda()
        obs, rew, done = eng.step(a)
        oobs, orew, odone = ora.step(own[t])
        o0, r0, d0 = ref0.step(base[t].reshape(2, 1))
        assert np.array_equal(obs[:, idx].cpu().numpy(), oobs), t
        assert np.array_equal(rew[:, idx].cpu().numpy(), orew), t
        rest = obs[:, mask]
        assert bool((rest == torch.from_numpy(o0).cuda()).all()), t
        assert bool((rew[:, mask] == int(r0[0, 0])).all()), t
        assert int(done.sum()) == (n if t in (199, 399) else 0), t
    eng.close()


def test_episode_control_masked_reset_and_stagger(layout_params):
    """f4 plumbing: device-side masked reset and per-kitchen timesteps (staggered horizons) vs the oracle."""
    params = layout_params["random1"]
    n, T = 997, 260
    acts = _mixed_actions("random1", T, n, seed=31)
    ora = oracle.OracleVecEnv(params, n, nthreads=4)
    eng = _engine(params, n)
    dacts = torch.from_numpy(acts).cuda()
    ts0 = eng.stagger().cpu().numpy().astype(np.int32)          # kitchen k starts at phase k*200//n
    ora.timestep[:] = ts0
    assert ts0.min() == 0 and ts0.max() == 199
    rng = np.random.default_rng(1)
    dones_per_step = []
    for t in range(T):
        if t in (40, 41, 150):
            mask = (rng.random(n) < 0.3).astype(np.int32)
            eng.episode_control(reset_mask=torch.from_numpy(mask))
            tiles, players, ts = ora.get_state()
            fresh = oracle.OracleVecEnv(params, n)
            ft, fp, fts = fresh.get_state()
            sel = mask.astype(bool)
            tiles[sel], players[sel], ts[sel] = ft[sel], fp[sel], fts[sel]
            ora.set_state(tiles, players, ts)
            assert np.array_equal(eng.observe().cpu().numpy(), ora.encode())
        obs, rew, done = eng.step(dacts[t])
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), oobs), t
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), t
        dones_per_step.append(int(odone.sum()))
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    assert max(dones_per_step[:39]) <= 6 and sum(dones_per_step) > n // 2   # resets arrive a few at a time, every step
    eng.close()


@pytest.mark.parametrize("layout,n", [("simple", 1000), ("random1", 517), ("unident_s", 300), ("six_pots", 257)])
def test_fused_rollout_equals_stepwise(layout, n, layout_params):
    """oc_rollout_fused (whole rollout in one launch, state kept on chip) vs the oracle stepped on the same
    actions: action tape (cyclic, int64) into time-major buffers, then the built-in counter-hash stream."""
    from diverse_conventions_amd import OvercookedEngine
    params = layout_params[layout]
    T = 215
    tape = _mixed_actions(layout, 60, n, seed=3)                     # shorter than T: used cyclically
    ora = oracle.OracleVecEnv(params, n, nthreads=4)
    eng = _engine(params, n)
    obs = torch.zeros((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
    rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    eng.rollout_fused(T, actions=torch.from_numpy(tape).cuda().to(torch.int64), obs=obs, reward=rew, done=done)
    for t in range(T):
        oobs, orew, odone = ora.step(tape[t % 60])
        assert np.array_equal(obs[t].cpu().numpy(), oobs), t
        assert np.array_equal(rew[t].cpu().numpy(), orew) and np.array_equal(done[t].cpu().numpy(), odone), t
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    # hash stream, outputs overwritten in place, continuing from the state above with a step offset
    acts = OvercookedEngine.synthetic_actions(99, n, 1000, 37)
    eng.rollout_fused(37, seed=99, t0=1000)
    for t in range(37):
        oobs, orew, odone = ora.step(acts[t])
    assert np.array_equal(eng.obs.cpu().numpy(), oobs) and np.array_equal(eng.reward.cpu().numpy(), orew)
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    # and ordinary steps continue seamlessly afterwards
    o2, r2, d2 = eng.step(torch.from_numpy(tape[0]).cuda())
    oo, orr, od = ora.step(tape[0])
    assert np.array_equal(o2.cpu().numpy(), oo) and np.array_equal(r2.cpu().numpy(), orr)
    eng.close()


def test_no_stray_writes_around_any_output(layout_params):
    """Every output lives inside a larger canary-filled allocation; after stepwise, C-loop and fused rollouts of a
    ragged kitchen count only the exact output ranges may have changed (GPU sanitizers are unavailable on the pool)."""
    params = layout_params["random0"]                 # S = 25: unpadded records are not 16-byte multiples
    n, pad = 1237, 4096
    eng = _engine(params, n)
    acts = torch.from_numpy(_mixed_actions("random0", 40, n, seed=6)).cuda()

    def guarded(shape, dtype, fill):
        numel = int(np.prod(shape))
        raw = torch.full((numel + 2 * pad,), fill, dtype=dtype, device="cuda")
        return raw, raw[pad:pad + numel].view(shape)

    raw_o, obs = guarded((3,) + tuple(eng.obs.shape), torch.int8, 0x55)
    raw_r, rew = guarded((3, 2, n), torch.int32, 0x55555555)
    raw_d, done = guarded((3, n), torch.int32, 0x55555555)
    eng.step(acts[0], obs=obs[0], reward=rew[0], done=done[0])
    eng.rollout(acts, 3, obs=obs, reward=rew, done=done)
    eng.rollout_fused(3, actions=acts, obs=obs, reward=rew, done=done)
    eng.rollout_fused(3, seed=5, obs=obs, reward=rew, done=done)
    torch.cuda.synchronize()
    for raw, fill in ((raw_o, 0x55), (raw_r, 0x55555555), (raw_d, 0x55555555)):
        assert bool((raw[:pad] == fill).all()) and bool((raw[-pad:] == fill).all())
    assert int((obs == 0x55).sum()) == 0               # and every byte inside was written (values are 0..20)
    eng.close()


def test_full_size_soak_every_kitchen(layout_params):
    """65 536 kitchens x 2000 steps with per-kitchen perturbed guided streams (cooking, deliveries, ten horizon
    resets): rewards and dones of EVERY kitchen compared every step, the full observation tensor every 97 steps,
    the full state at the end -- no sampling."""
    params = layout_params["simple"]
    n, T = 65536, 2000
    g = load_golden(os.path.join(GOLDEN, "guided_simple.npz"))["actions"]          # 600 steps, resets in sync
    rng = np.random.default_rng(123)
    ora = oracle.OracleVecEnv(params, n, nthreads=16)
    eng = _engine(params, n)
    total = 0
    for t in range(T):
        base = np.broadcast_to(g[t % 600][:, None], (2, n))
        acts = np.where(rng.random((1, n)) < 0.9, base, rng.integers(0, 6, size=(2, n))).astype(np.int32)
        obs, rew, done = eng.step(torch.from_numpy(acts).cuda())
        want_obs = t % 97 == 0 or t == T - 1
        oobs, orew, odone = ora.step(acts, want_obs=want_obs)
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), t
        if want_obs:
            assert np.array_equal(obs.cpu().numpy(), oobs), t
        total += int(orew[0].sum())
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    assert total > 10 * n                       # the streams keep cooking and delivering at this scale too
    eng.close()


def test_env_checkpoint_roundtrip(layout_params, tmp_path):
    """state_dict -> torch.save -> torch.load -> load_state_dict into a fresh engine: the rollout continues
    exactly where it stopped."""
    params = layout_params["random3"]
    n = 333
    acts = torch.from_numpy(_mixed_actions("random3", 120, n, seed=8)).cuda()
    a = _engine(params, n)
    a.rollout(acts, 77)
    torch.save(a.state_dict(), tmp_path / "env.pt")
    b = _engine(params, n)
    b.load_state_dict(torch.load(tmp_path / "env.pt", weights_only=False))
    assert torch.equal(a.observe(), b.obs)
    for t in range(77, 120):
        a.step(acts[t]); b.step(acts[t])
    assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward)
    other = _engine(layout_params["simple"], n)
    with pytest.raises(ValueError):
        other.load_state_dict(a.state_dict())
    # corrupted checkpoints are refused on the device, all or nothing: a validating pass that writes nothing
    # (oc_validate_states) counts the damaged records, and the engine keeps the state it had
    from diverse_conventions_amd.native import OvercookedNativeError
    sd = a.state_dict()
    good = sd["records"].clone()
    S = 40
    sd["records"][5, S] = 0                        # player 0 on tile 0: a wall
    sd["records"][9, S] = int(good[9, S]) | (7 << 8)          # orientation 7
    sd["records"][200, 3] = 9                      # object code 9 on a tile
    sd["records"][201, S + 1] = 77                 # ... and in a hand
    c = _engine(params, n)
    c.rollout(acts, 5)
    before = c.export_states().clone()
    with pytest.raises(OvercookedNativeError, match="4 checkpoint record"):
        c.load_state_dict(sd)
    assert torch.equal(c.export_states(), before)              # nothing of the bad checkpoint was loaded
    # the record-by-record import (oc_import_states_checked) skips exactly the damaged ones
    assert c.import_states_checked(sd["records"].cuda().to(torch.int32)) == 4
    after = c.export_states().cpu()
    bad = [5, 9, 200, 201]
    keep = np.setdiff1d(np.arange(n), bad)
    assert torch.equal(after[bad], before.cpu()[bad]) and torch.equal(after[keep], good[keep].to(torch.int32))
    for e in (a, b, other, c):
        e.close()


def test_create_destroy_does_not_leak(layout_params):
    """50 engines of 100k kitchens created, stepped and closed: device memory returns to where it was."""
    params = layout_params["unident_s"]
    torch.cuda.synchronize()
    warm = _engine(params, 100_000); warm.close()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(50):
        eng = _engine(params, 100_000)
        eng.step(torch.zeros((2, 100_000), dtype=torch.int32, device="cuda"))
        if i % 10 == 0:
            eng.set_tuning(8 + 4 * (i // 10), 2, 8)      # re-specialises: the old code object must be unloaded
        eng.close()
        del eng
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 * 2**20, (free0, free1)


def test_hip_replays_the_reference_syncvectorenv_recording(layout_params):
    """The HIP engine directly against tests/golden/syncvec_simple.npz -- the reference's own SyncVectorEnv
    (pantheonrl_extension/vectorenv.py:362-396) over 3 envs for 420 steps: the batched surface with its auto-reset rule
    (terminal reward and done kept, the reset observation returned), six resets in the recording.  Also through the
    drop-in class, whose ego view must be the recording's player-0 rows."""
    from diverse_conventions_amd import OvercookedMadrona, RandomVectorAgent
    rec = load_golden(os.path.join(GOLDEN, "syncvec_simple.npz"))
    T = rec["actions"].shape[0]
    eng = _engine(layout_params["simple"], 3)
    assert np.array_equal(eng.observe().cpu().numpy(), rec["init_obs"])
    acts = torch.from_numpy(rec["actions"].astype(np.int32)).cuda()
    for t in range(T):
        obs, rew, done = eng.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), rec["obs"][t]), t
        assert np.array_equal(rew.cpu().numpy(), rec["reward"][t]), t
        assert np.array_equal(done.cpu().numpy().astype(np.uint8), rec["done"][t]), t
    eng.close()
    assert rec["done"].sum() == 6
    env = OvercookedMadrona("simple", 3, 0, debug_compile=False)
    step_no = [0]
    env.add_partner_agent(RandomVectorAgent(lambda: acts[step_no[0], 1].unsqueeze(-1)))
    vobs = env.reset()
    assert np.array_equal(vobs.obs.cpu().numpy(), rec["init_obs"][0])
    for t in range(T):
        step_no[0] = t
        vobs, rew, done, _ = env.step(acts[t, 0].unsqueeze(-1))
        assert np.array_equal(vobs.obs.cpu().numpy(), rec["obs"][t][0]), t
        assert np.array_equal(rew.cpu().numpy(), rec["reward"][t][0]) and np.array_equal(done.cpu().numpy().astype(np.uint8), rec["done"][t])
    env.close()


def test_this_whole_file_on_the_generic_ahead_of_time_kernel(clean_launcher):
    """A box where hiprtc is unusable serves every engine with the generic ahead-of-time kernel (every layout parameter a runtime
    field).  The driver never sets OC_NO_JIT, so this test does: the whole parity file again, in a child process, on that kernel."""
    import os
    import sys
    if os.environ.get("OC_NO_JIT", "0") not in ("", "0"):
        pytest.skip("already running on the generic kernel")
    r = clean_launcher.run([sys.executable, "-m", "pytest", "tests/test_hip_parity.py", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                           env={"OC_NO_JIT": "1"}, timeout=1100)
    tail = r["stdout"][-1500:] + r["stderr"][-500:]
    assert r["rc"] == 0 and " passed" in r["stdout"] and "failed" not in r["stdout"], tail


def test_per_xcd_group_sizes_with_a_halo_kitchen_on_the_diagnostics_build(clean_launcher):
    """The experiment of round 4: groups sized per XCD, kitchen by kitchen, their plane bytes cut at multiples of
    16 tiles with the neighbour's last kitchen stepped as a halo (oc_step_body.h, OC_XW_SUM).  Not a product path -- it gained nothing --
    but it shares the body's group geometry with every kernel, so its parity is kept: every kitchen of a ragged batch against the oracle,
    in place and into time-major slots with records, group sizes that include 0 and 1 (tools/check_xcc_weights.py, csrc/liboc_hip_diag.so)."""
    import os
    import sys
    from diverse_conventions_amd import native
    diag = os.path.join(native.CSRC, "liboc_hip_diag.so")
    newest = max(os.path.getmtime(os.path.join(native.CSRC, f)) for f in os.listdir(native.CSRC) if f.endswith((".hip", ".h")))
    if not os.path.exists(diag) or os.path.getmtime(diag) < newest:
        pytest.skip("csrc/liboc_hip_diag.so is not built for these sources (OC_DIAGNOSTICS=1 python -c 'from diverse_conventions_amd import native; native.build()')")
    for argv in (["cramped_room", "8191", "24", "19,13,17,15,16,0,23,1"], ["coordination_ring", "6001", "20", "17,15,17,15,17,15,17,15"]):
        r = clean_launcher.run([sys.executable, "tools/check_xcc_weights.py"] + argv, env={"OC_DIAGNOSTICS": "1"}, timeout=600)
        assert r["rc"] == 0 and "bit-exact" in r["stdout"], r["stdout"][-1500:] + r["stderr"][-800:]


@pytest.mark.parametrize("layout,n", [("cramped_room", 49153), ("cramped_room", 49154), ("cramped_room", 49155), ("asymmetric_advantages", 49157)])
def test_time_major_steps_when_the_viewer_planes_start_off_64_byte_boundaries(layout, n):
    """Kitchen counts whose planes are no multiple of 64 bytes: the second viewer's plane, and every other time-major slot, then start
    16 / 32 / 48 bytes off a boundary, and the module for fresh destinations owns each plane's bytes in ranges that start a few tiles
    early, the neighbour's last kitchen stepped as a halo (oc_step_body.h, OC_HALO_R).  Every kitchen against the oracle, into
    consecutive slots (both slot parities), records included, in place in between."""
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
    from oracle import oracle
    params = get_base_layout_params(layout, 9)
    eng = OvercookedEngine(params, n, 0)
    ora = oracle.OracleVecEnv(params, n)
    W, H = params["width"], params["height"]
    T = 14
    slots = torch.zeros((T, 2, n, W, H, 20), dtype=torch.int8, device="cuda")
    rew = torch.zeros((T, 2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
    recs = torch.zeros((T, n, eng.export_states().shape[-1]), dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(n)
    for t in range(T):
        a = torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32, generator=g)
        if t % 5 == 4:
            obs, r, d = eng.step(a)
        else:
            obs, r, d = eng.step(a, obs=slots[t], reward=rew[t], done=done[t], records=recs[t])
        oobs, orew, odone = ora.step(a.cpu().numpy())
        assert np.array_equal(obs.cpu().numpy(), oobs), t
        assert np.array_equal(r.cpu().numpy(), orew) and np.array_equal(d.cpu().numpy().astype(bool), np.asarray(odone).astype(bool)), t
        if t % 5 != 4:
            assert torch.equal(recs[t], eng.export_states()), t
    eng.close()


@pytest.mark.parametrize("cus", [2, 0])
def test_halo_kitchens_are_read_from_the_block_the_launch_does_not_write(cus, clean_launcher):
    """The wavefront that owns a halo kitchen writes its post-step record in the same launch in which its neighbour reads the
    pre-step one: such launches read one record block and write the other (halo_state_out in overcooked_engine.hip), so the order in
    which wavefronts run cannot matter.  Here they run in very different orders: 131 075 kitchens = 8193 groups on the 16 persistent
    wavefronts of two CUs (512 groups each, a wavefront's later groups long after its neighbours' earlier ones) and on the whole chip;
    every kitchen against the oracle, records included, with in-place steps (which do not swap the blocks) in between."""
    import os
    import sys
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params, native
    if cus and not native.DIAGNOSTICS:
        # the CU budget is an experiment's knob (oc_exp_set_cu_budget): since round 6 only the diagnostics library exports it, so the
        # two-CU case runs this very test as a child process on that build (same kernel text, same halo path)
        diag = os.path.join(native.CSRC, "liboc_hip_diag.so")
        newest = max(os.path.getmtime(os.path.join(native.CSRC, f)) for f in os.listdir(native.CSRC) if f.endswith((".hip", ".h")))
        if not os.path.exists(diag) or os.path.getmtime(diag) < newest:
            pytest.skip("csrc/liboc_hip_diag.so is not built for these sources")
        r = clean_launcher.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.abspath(__file__),
                                "-k", "test_halo_kitchens_are_read_from_the_block_the_launch_does_not_write and 2"],
                               env={"OC_DIAGNOSTICS": "1"}, timeout=900)
        assert r["rc"] == 0 and "1 passed" in r["stdout"], r["stdout"][-1500:] + r["stderr"][-500:]
        return
    params = get_base_layout_params("cramped_room", 7)
    n, T = 131075, 9
    eng = OvercookedEngine(params, n, 0)
    if cus:
        native.check(native.lib().oc_exp_set_cu_budget(eng._handle, cus))
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    slots = torch.zeros((3, 2, n, params["width"], params["height"], 20), dtype=torch.int8, device="cuda")
    rew = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros(n, dtype=torch.int32, device="cuda")
    recs = torch.zeros((n, eng.export_states().shape[-1]), dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(T):
        a = torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32, generator=g)
        if t in (3, 6):
            obs, r, d = eng.step(a)
        else:
            obs, r, d = eng.step(a, obs=slots[t % 3], reward=rew, done=done, records=recs)
        oobs, orew, odone = ora.step(a.cpu().numpy())
        assert np.array_equal(obs.cpu().numpy(), oobs), t
        assert np.array_equal(r.cpu().numpy(), orew) and np.array_equal(d.cpu().numpy().astype(bool), np.asarray(odone).astype(bool)), t
        if t not in (3, 6):
            assert torch.equal(recs, eng.export_states()), t
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()


def test_a_captured_step_into_a_fresh_slot_replays_without_halo_kitchens():
    """A launch captured into a graph cannot swap record blocks on replay, so it runs without halo kitchens (state_out NULL): the same
    graph replayed three times against the oracle, at a kitchen count whose planes start off 64-byte boundaries."""
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
    params = get_base_layout_params("cramped_room", 50)
    n = 49155
    eng = OvercookedEngine(params, n, 0)
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    eng.prepare_kernels()
    a = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    slots = torch.zeros((2, 2, n, params["width"], params["height"], 20), dtype=torch.int8, device="cuda")
    rew = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros(n, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(9)
    draw = lambda: torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32, generator=g)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(2):                                   # (eager: these launches have halo kitchens and swap the blocks)
            a.copy_(draw())
            eng.step(a, obs=slots[i % 2], reward=rew, done=done)
            ora.step(a.cpu().numpy())
    torch.cuda.current_stream().wait_stream(s)
    a.copy_(draw())
    host_a = a.cpu().numpy()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.step(a, obs=slots[0], reward=rew, done=done)
    for i in range(3):
        if i:
            a.copy_(draw())
            host_a = a.cpu().numpy()
        graph.replay()
        oobs, orew, odone = ora.step(host_a)
        assert np.array_equal(slots[0].cpu().numpy(), oobs), i
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy().astype(bool), np.asarray(odone).astype(bool)), i
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()


def test_capture_then_an_eager_step_into_a_fresh_slot_then_replay():
    """The advisor's round-5 sequence: a graph is captured (its kernel arguments hold the record block's address), then an EAGER step goes
    into a fresh destination -- the launch that used to swap the record blocks -- then the graph is replayed.  From the first capture on
    the engine's blocks never swap (note_stream / halo_state_out in overcooked_engine.hip: such launches then run without halo kitchens),
    so graph and eager calls keep stepping the same kitchens: every replay, every eager step and the final records against the oracle."""
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
    params = get_base_layout_params("cramped_room", 30)
    n = 49155                                                # planes start off 64-byte boundaries: the module for fresh destinations has halo kitchens
    eng = OvercookedEngine(params, n, 0)
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    eng.prepare_kernels()
    a = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    slots = torch.zeros((3, 2, n, params["width"], params["height"], 20), dtype=torch.int8, device="cuda")
    rew = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    done = torch.zeros(n, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(11)

    def draw():
        a.copy_(torch.randint(0, 6, (2, n), device="cuda", dtype=torch.int32, generator=g))
        return a.cpu().numpy()

    def check(obs, where):
        oobs, orew, odone = ora.step(host_a)
        assert np.array_equal(obs.cpu().numpy(), oobs), where
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy().astype(bool), np.asarray(odone).astype(bool)), where

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):                                   # eager, before any capture: halo launches that swap the blocks (an odd number: the blocks end up swapped)
            host_a = draw()
            eng.step(a, obs=slots[i % 2], reward=rew, done=done)
            check(slots[i % 2], ("eager before", i))
    torch.cuda.current_stream().wait_stream(s)
    host_a = draw()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.step(a, obs=slots[0], reward=rew, done=done)    # (capture enqueues nothing: the oracle steps at the replays)
    for rnd in range(3):
        host_a = draw()
        graph.replay()
        check(slots[0], ("replay", rnd))
        for j in (1, 2):                                     # eager steps into destinations that differ from the previous step's: fresh slots
            host_a = draw()
            eng.step(a, obs=slots[j], reward=rew, done=done)
            check(slots[j], ("eager after capture", rnd, j))
        assert torch.equal(eng.export_states(), eng.export_states())
    host_a = draw()
    graph.replay()
    check(slots[0], "last replay")
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()


@pytest.mark.parametrize("layout,n,dtype", [("cramped_room", 4099, torch.int64), ("asymmetric_advantages", 8192, torch.int32), ("coordination_ring", 65536, torch.int64)])
def test_step_from_two_action_buffers_equals_the_stacked_step(layout, n, dtype):
    """oc_step_players (round 6): the two players' actions as two [N] buffers -- what VectorMultiAgentEnv.step holds before it stacks
    them -- against the oracle, and against oc_step on the stacked tensor from the same states; falls back (False) for a destination
    that is not the previous step's, for mixed dtypes and for non-contiguous tensors."""
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
    params = get_base_layout_params(layout, 30)
    eng, twin = OvercookedEngine(params, n, 0), OvercookedEngine(params, n, 0)
    if "specialised" not in eng.kernel_build_info:           # (this file's second run, OC_NO_JIT=1: the generic kernel has no two-buffer entry point)
        z = torch.zeros(n, dtype=dtype, device="cuda")
        assert eng.step_players([z, z]) is False
        eng.close(); twin.close()
        return
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    g = torch.Generator(device="cuda").manual_seed(n)
    for t in range(70):                                      # crosses the horizon twice
        a0 = torch.randint(0, 6, (n, 1), device="cuda", dtype=dtype, generator=g)
        a1 = torch.randint(0, 6, (n,), device="cuda", dtype=dtype, generator=g)
        assert eng.step_players([a0, a1]) is True, t
        twin.step(torch.stack([a0.view(-1), a1]))
        oobs, orew, odone = ora.step(torch.stack([a0.view(-1), a1]).cpu().numpy())
        assert np.array_equal(eng.obs.cpu().numpy(), oobs), t
        assert np.array_equal(eng.reward.cpu().numpy(), orew) and np.array_equal(eng.done.cpu().numpy().astype(bool), np.asarray(odone).astype(bool)), t
        assert torch.equal(eng.obs, twin.obs) and torch.equal(eng.reward, twin.reward) and torch.equal(eng.export_states(), twin.export_states())
    a0 = torch.randint(0, 6, (n,), device="cuda", dtype=dtype, generator=g)
    a1 = torch.randint(0, 6, (n,), device="cuda", dtype=dtype, generator=g)
    assert eng.step_players([a0, a1.to(torch.int32 if dtype == torch.int64 else torch.int64)]) is False      # mixed dtypes
    assert eng.step_players([a0, torch.stack([a1, a1], dim=1)[:, 0]]) is False                               # not contiguous
    slot = torch.zeros_like(eng.obs)
    eng.step(torch.stack([a0, a1]), obs=slot)                # a step into another destination ...
    assert eng.step_players([a0, a1]) is False               # ... and the next one into the engine's own buffer is not "in place": the caller stacks
    eng.step(torch.stack([a0, a1]))
    assert eng.step_players([a0, a1]) is True
    eng.close(); twin.close()
