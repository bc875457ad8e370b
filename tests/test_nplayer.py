"""Kitchens with 1, 3 and 4 players (the reference's Python MDP and encoder are N-player: envs/overcooked2_reimplement.py
:221-234, :261-263; envs/overcooked2_env.py:29 `num_players`; layouts multiplayer_schelling, simple_single).  The HIP engine
runs them on the group kernel of csrc/oc_np_kernels.h (one launch per step since round 5); parity bar as everywhere: bit-exact against fixtures recorded from the
reference (tests/golden/np_*.npz, make_golden_nplayer.py) and against the oracle on batches."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_files, load_golden
from oracle import oracle

pytestmark = pytest.mark.gpu

NP_FIXTURES = golden_files("np_random_") + golden_files("np_guided_")


def nplayer_params():
    with open(os.path.join(GOLDEN, "layout_params_nplayer.json")) as f:
        return json.load(f)


def case_of(path):
    return os.path.basename(path)[:-4].split("_", 2)[2]


@pytest.mark.parametrize("path", NP_FIXTURES, ids=[os.path.basename(p)[:-4] for p in NP_FIXTURES])
def test_hip_replays_reference_nplayer_fixture(path):
    from diverse_conventions_amd import OvercookedEngine
    rec = load_golden(path)
    params = nplayer_params()[case_of(path)]
    P = params["num_players"]
    eng = OvercookedEngine(params, 1, 0)
    assert tuple(eng.obs.shape) == (P, 1, params["width"], params["height"], 5 * P + 10) and f"{P}-player" in eng.kernel_build_info
    eng.set_state(rec["init_tiles"][None], rec["init_players"][None], rec["init_timestep"][None])
    assert np.array_equal(eng.observe().cpu().numpy()[:, 0], rec["init_obs"])
    T = len(rec["actions"])
    acts = torch.from_numpy(rec["actions"].astype(np.int32)).cuda().reshape(T, P, 1)
    for t in range(T):
        obs, rew, done = eng.step(acts[t])
        assert np.array_equal(obs.cpu().numpy()[:, 0], rec["obs"][t]), f"step {t}: obs"
        assert np.all(rew.cpu().numpy()[:, 0] == rec["reward"][t]) and int(done.cpu()[0]) == int(rec["done"][t]), f"step {t}"
        tiles, players, ts = eng.get_state()
        assert np.array_equal(tiles[0], rec["tiles"][t]) and np.array_equal(players[0], rec["players"][t]) and int(ts[0]) == int(rec["timestep"][t])
    eng.close()


@pytest.mark.parametrize("case", ["schelling4", "schelling3", "single"])
def test_hip_vs_oracle_nplayer_batch(case):
    """1003 kitchens (ragged blocks), 430 steps over two horizon resets, per-kitchen streams built from the fixture's guided
    stream with random substitutions; int64 [P, N, 1] actions on odd steps."""
    from diverse_conventions_amd import OvercookedEngine
    params = nplayer_params()[case]
    P, n, T = params["num_players"], 1003, 430
    g = load_golden(os.path.join(GOLDEN, f"np_guided_{case}.npz"))["actions"]
    rng = np.random.default_rng(4)
    base = np.broadcast_to(g[:T, :, None], (T, P, n))
    acts = np.where(rng.random((T, 1, n)) < 0.85, base, rng.integers(0, 6, size=(T, P, n))).astype(np.int32)
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    eng = OvercookedEngine(params, n, 0)
    assert np.array_equal(eng.observe().cpu().numpy(), ora.encode())
    dacts = torch.from_numpy(acts).cuda()
    total = 0
    for t in range(T):
        a = dacts[t].to(torch.int64).unsqueeze(-1) if t % 2 else dacts[t]
        obs, rew, done = eng.step(a)
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), f"step {t}"
        assert np.array_equal(obs.cpu().numpy(), oobs), f"step {t}: obs"
        total += int(orew[0].sum())
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    assert total > 0 and int(ora.timestep[0]) == T - 400
    # checkpoint reload for this player count (state_dict -> load_state_dict through the device-side validating pass)
    twin = OvercookedEngine(params, n, 0)
    twin.load_state_dict(eng.state_dict())
    assert torch.equal(twin.obs, eng.observe())
    eng.step(dacts[0]); twin.step(dacts[0])
    assert torch.equal(twin.obs, eng.obs) and torch.equal(twin.export_states(), eng.export_states())
    eng.close(); twin.close()


@pytest.mark.parametrize("case,n", [("schelling4", 36001), ("schelling3", 33001), ("single", 70003)])
def test_nplayer_large_batches_and_launch_geometries(case, n):
    """Batches large enough for the 16-kitchen groups of the N-player kernel (persistent wavefronts walk several groups, the next
    group's records prefetched), odd kitchen counts (ragged last group; with three players every viewer plane but the first starts
    off a dword: 1..3 single bytes, then 16-byte stores of LDS reads at that byte offset), against the oracle with records; then the
    same steps under other launch geometries (oc_set_tuning: kitchens per group, passes per flush, wavefronts per CU) must give
    identical bytes."""
    from diverse_conventions_amd import OvercookedEngine
    params = dict(nplayer_params()[case], horizon=7)
    P, T = params["num_players"], 10
    rng = np.random.default_rng(n)
    acts = rng.integers(0, 6, size=(T, P, n)).astype(np.int32)
    acts[:, :, ::3] = 5                                                # a third of the kitchens interact all the time
    dacts = torch.from_numpy(acts).cuda()
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    eng = OvercookedEngine(params, n, 0)
    recs = torch.zeros((T, n, eng.record_bytes // 4), dtype=torch.int32, device="cuda")
    want_obs = []
    for t in range(T):
        obs, rew, done = eng.step(dacts[t], records=recs[t])
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), oobs), t
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), t
        assert torch.equal(recs[t], eng.export_states()), t
        want_obs.append(obs.clone())
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()
    # the same T steps as ONE launch (oc_rollout_fused: records resident in LDS between the steps), time-major outputs with records
    e1 = OvercookedEngine(params, n, 0)
    o1 = torch.zeros((T,) + tuple(e1.obs.shape), dtype=torch.int8, device="cuda")
    r1, d1 = torch.zeros((T, P, n), dtype=torch.int32, device="cuda"), torch.zeros((T, n), dtype=torch.int32, device="cuda")
    rec1 = torch.zeros_like(recs)
    e1.rollout_fused(T, actions=dacts, obs=o1, reward=r1, done=d1, records=rec1)
    assert torch.equal(o1, torch.stack(want_obs)) and torch.equal(rec1, recs) and torch.equal(e1.export_states(), recs[T - 1])
    assert int(d1.sum()) > 0                                            # the horizon was crossed inside the launch
    # ... and in place with a tape shorter than the rollout (slices reused modulo its length)
    e1.reset_all()
    e1.rollout_fused(T, actions=dacts[:3].contiguous())
    e3 = OvercookedEngine(params, n, 0)
    for t in range(T):
        e3.step(dacts[t % 3])
    assert torch.equal(e1.obs, e3.obs) and torch.equal(e1.reward, e3.reward) and torch.equal(e1.done, e3.done) and torch.equal(e1.export_states(), e3.export_states())
    e1.close(); e3.close()
    recq = eng.record_bytes // 16                                      # (a group's records fit the prefetch window: K x recq <= 256 chunks)
    for K, G, w in ((16 if P > 1 else 32, 3, 8), (5, 1, 3), (min(64 // P, 256 // recq), 4, 2), (1, 2, 32)):
        e2 = OvercookedEngine(params, n, 0)
        e2.set_tuning(K, G, w)
        for t in range(T):
            obs, _, _ = e2.step(dacts[t])
            assert torch.equal(obs, want_obs[t]), (K, G, w, t)
        assert torch.equal(e2.export_states(), recs[T - 1])
        e2.close()


@pytest.mark.parametrize("case,n", [("schelling3", 1003), ("schelling4", 5000), ("single", 777)])
def test_nplayer_trajectory_featurised_by_one_launch(case, n):
    """oc_encode_trajectory with 1, 3, 4 players: ONE launch of the group kernel over all T blocks of records (a launch per time step
    before round 5) -- strided record blocks (every other slot of a larger tensor), written at a kitchen offset inside a wider
    [T, P, N_total, W, H, C] tensor whose planes then start off 16-byte (three players: off dword) boundaries, the rest of the tensor
    untouched; against per-step oc_encode_states of the same records, and one block that is only dword aligned."""
    from diverse_conventions_amd import OvercookedEngine
    params = dict(nplayer_params()[case], horizon=9)
    P, T = params["num_players"], 7
    eng = OvercookedEngine(params, n, 0)
    g = torch.Generator(device="cuda").manual_seed(n)
    big = torch.zeros((2 * T, n, eng.record_bytes // 4), dtype=torch.int32, device="cuda")
    for t in range(T):
        eng.step(torch.randint(0, 6, (P, n), device="cuda", dtype=torch.int32, generator=g), records=big[2 * t])
    recs = big[::2]                                                     # [T, n, rec/4], block stride 2 x n x rec
    want = torch.stack([eng.encode_states(recs[t].contiguous()) for t in range(T)])            # [T, P, n, W, H, C]
    off, n_total = 3, n + 8
    out = torch.full((T, P, n_total) + tuple(want.shape[3:]), 77, dtype=torch.int8, device="cuda")
    eng.encode_trajectory_into(recs, out, off)
    assert torch.equal(out[:, :, off:off + n], want)
    assert bool((out[:, :, :off] == 77).all()) and bool((out[:, :, off + n:] == 77).all())
    # a record block that starts 4 bytes off a 16-byte boundary (the pipeline's 16-byte loads do not apply: dword loads)
    flat = torch.zeros(recs[0].numel() + 1, dtype=torch.int32, device="cuda")
    flat[1:] = recs[T - 1].reshape(-1)
    assert flat[1:].data_ptr() % 16 == 4
    one = torch.zeros((1, P, n) + tuple(want.shape[3:]), dtype=torch.int8, device="cuda")
    eng.encode_trajectory_into(flat[1:].view(1, n, -1), one, 0)
    assert torch.equal(one[0], want[T - 1])
    eng.close()


def test_four_player_drop_in_env_and_buffers():
    """OvercookedMadrona("multiplayer_schelling", ...) as the reference constructs it (envs/overcooked2_env.py:29): three partner
    agents, observations (N, 7, 7, 30); the time-major TrajectoryBuffer, per-kitchen episode control and state export / import /
    re-featurisation work for any player count, and so do tape rollouts, groups and trajectory re-featurisation (since round 5
    one launch each, groups a launch per engine); only the built-in 2-player action stream refuses politely."""
    from diverse_conventions_amd import (EngineGroup, OvercookedEngine, OvercookedMadrona, RandomVectorAgent, TrajectoryBuffer,
                                         get_base_layout_params)
    from diverse_conventions_amd.native import OvercookedNativeError
    params = nplayer_params()["schelling4"]
    n, T = 64, 30
    rng = np.random.default_rng(1)
    acts = rng.integers(0, 6, size=(T, 4, n, 1))
    dacts = torch.from_numpy(acts).cuda()
    env = OvercookedMadrona("multiplayer_schelling", n, 0, debug_compile=False)
    assert env.n_players == 4 and tuple(env.observation_space.shape) == (7, 7, 30)
    step_no = [0]
    for pl in (1, 2, 3):
        env.add_partner_agent(RandomVectorAgent(lambda pl=pl: dacts[step_no[0], pl]), player_num=pl)
    ora = oracle.OracleVecEnv(params, n)
    vobs = env.reset()
    assert vobs.obs.shape == (n, 7, 7, 30) and np.array_equal(vobs.obs.cpu().numpy(), ora.encode()[0])
    for t in range(T):
        step_no[0] = t
        vobs, rew, done, _ = env.step(dacts[t, 0])
        oobs, orew, odone = ora.step(acts[t, :, :, 0])
        assert np.array_equal(vobs.obs.cpu().numpy(), oobs[0]) and np.array_equal(rew.cpu().numpy(), orew[0])
    env.close()
    eng = OvercookedEngine(params, n, 0)
    buf = TrajectoryBuffer(eng, T, store_states=True)
    ora = oracle.OracleVecEnv(params, n)
    for t in range(T):
        buf.step(dacts[t])
        ora.step(acts[t, :, :, 0])
    assert buf.obs.shape == (T + 1, 4, n, 7, 7, 30) and np.array_equal(buf.obs[T].cpu().numpy(), ora.encode())
    assert np.array_equal(eng.encode_states(buf.records[T]).cpu().numpy(), ora.encode())       # re-featurised from the compact records
    # an odd number of 3-player kitchens on a 7x7 grid puts the second viewer's plane at an odd byte offset (n * 49 * 25): the
    # N-player encoder takes it (byte stores where a block's destination is not dword aligned), as oc_step / oc_observe always did
    p3 = nplayer_params()["schelling3"]
    e3 = OvercookedEngine(p3, 5, 0)
    o3 = oracle.OracleVecEnv(p3, 5)
    a3 = rng.integers(0, 6, size=(6, 3, 5)).astype(np.int32)
    for t in range(6):
        e3.step(torch.from_numpy(a3[t]).cuda()); o3.step(a3[t])
    assert np.array_equal(e3.encode_states(e3.export_states()).cpu().numpy(), o3.encode())
    e3.close()
    other = OvercookedEngine(params, n, 0)
    other.import_states(eng.export_states())
    assert torch.equal(other.observe(), buf.obs[T])
    mask = torch.zeros(n, dtype=torch.int32, device="cuda"); mask[::2] = 1
    eng.episode_control(reset_mask=mask, timesteps=torch.full((n,), 195, dtype=torch.int32, device="cuda"))
    tiles, players, ts = eng.get_state()
    start = oracle.OracleVecEnv(params, 1).get_state()
    assert np.array_equal(tiles[0], start[0][0]) and np.array_equal(players[0], start[1][0]) and np.all(ts == 195)
    _, _, d = eng.rollout(dacts[:5, :, :, 0].to(torch.int32).contiguous(), 5)
    assert int(d.sum()) == n                                          # every kitchen reached the horizon at the fifth step
    # what is ONE fused launch for two players is one launch here as well (tape rollouts, trajectories; a launch per engine for groups):
    # a tape rollout (oc_rollout / oc_rollout_fused) against the oracle ...
    tape = rng.integers(0, 6, size=(7, 4, n)).astype(np.int32)
    dtape = torch.from_numpy(tape).cuda()
    ora2 = oracle.OracleVecEnv(params, n)
    for e, roll in ((eng, lambda e, o, r, d: e.rollout(dtape, 7, obs=o, reward=r, done=d)),
                    (other, lambda e, o, r, d: e.rollout_fused(7, actions=dtape, obs=o, reward=r, done=d))):
        e.reset_all()
        o = torch.zeros((7,) + tuple(e.obs.shape), dtype=torch.int8, device="cuda")
        r, d = torch.zeros((7, 4, n), dtype=torch.int32, device="cuda"), torch.zeros((7, n), dtype=torch.int32, device="cuda")
        roll(e, o, r, d)
        ora2.reset_all()
        for t in range(7):
            oobs, orew, odone = ora2.step(tape[t])
            assert np.array_equal(o[t].cpu().numpy(), oobs) and np.array_equal(r[t].cpu().numpy(), orew) and np.array_equal(d[t].cpu().numpy(), odone), t
    with pytest.raises(OvercookedNativeError, match="2-player"):
        eng.rollout_fused(3, seed=1)                                  # only the built-in action stream is defined for two players
    # ... a group that mixes player counts (one launch per engine inside, oc_group_step outside) ...
    two = OvercookedEngine(get_base_layout_params("simple", 200), 50, 0)
    two_solo = OvercookedEngine(get_base_layout_params("simple", 200), 50, 0)
    grp = EngineGroup([eng, two])
    assert "one launch per engine" in grp.build_info
    a2 = torch.from_numpy(rng.integers(0, 6, size=(2, 50)).astype(np.int32)).cuda()
    gobs, grew, gdone = grp.step([dtape[0], a2])
    other.step(dtape[0]); two_solo.step(a2)
    assert torch.equal(gobs[0], other.obs) and torch.equal(gobs[1], two_solo.obs) and torch.equal(grew[0], other.reward) and torch.equal(gdone[1], two_solo.done)
    grp.close(); two.close(); two_solo.close()
    # ... and a whole trajectory re-featurised from compact records (oc_encode_trajectory: T plain encode launches)
    from diverse_conventions_amd import TrajectoryExchange
    assert torch.equal(TrajectoryExchange().featurize_all(eng, buf.records.unsqueeze(0)), buf.obs)
    # checkpoint reload with other player counts: the device-side check knows the N-player record, a damaged checkpoint is
    # refused whole and a good one continues bit-exactly
    sd = eng.state_dict()
    fresh = OvercookedEngine(params, n, 0)
    fresh.load_state_dict(sd)
    assert torch.equal(fresh.export_states(), eng.export_states()) and torch.equal(fresh.obs, eng.observe())
    a5 = dacts[5:9, :, :, 0].to(torch.int32).contiguous()
    eng.rollout(a5, 4); fresh.rollout(a5, 4)
    assert torch.equal(fresh.obs, eng.obs) and torch.equal(fresh.export_states(), eng.export_states())
    bad = eng.state_dict()
    S, P = params["width"] * params["height"], params["num_players"]
    bad["records"][3, S + P - 1] = 0                                  # the last player on tile 0: a wall
    bad["records"][7, S + 2 * P - 1] = 9                              # object code 9 in the last player's hands
    before = fresh.export_states().clone()
    with pytest.raises(OvercookedNativeError, match="2 checkpoint record"):
        fresh.load_state_dict(bad)
    assert torch.equal(fresh.export_states(), before)
    assert fresh.import_states_checked(bad["records"].cuda().to(torch.int32)) == 2
    eng.close(); other.close(); fresh.close()
