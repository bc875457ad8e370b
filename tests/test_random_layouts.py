"""Differential tests on random kitchens nobody shipped: the two CPU oracles (each pinned to the reference on
its own) against each other, and the HIP engine against the C oracle -- grids up to 10x7, islands, several pots
and dispensers, non-uniform recipe tables, short horizons."""
import numpy as np
import pytest

from oracle import oracle
from oracle.py_oracle import PyKitchen
from random_layouts import busy_actions, busy_actions_players, random_layout, random_layout_players


@pytest.mark.parametrize("seed", range(24))
def test_c_oracle_equals_python_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    params = random_layout(rng)
    n, T = 3, 260
    acts = busy_actions(rng, T, n)
    env = oracle.OracleVecEnv(params, n)
    py = [PyKitchen(params) for _ in range(n)]
    total = 0
    for t in range(T):
        obs, rew, done = env.step(acts[t])
        for k in range(n):
            o, r, d = py[k].n_step([int(acts[t, 0, k]), int(acts[t, 1, k])])
            assert r == int(rew[0, k]) and d == bool(done[k]), (seed, t, k)
            assert np.array_equal(np.array(o, np.int8), obs[:, k]), (seed, t, k)
        total += int(rew[0].sum())
    tiles, players, ts = env.get_state()
    for k in range(n):
        pt, pp, pts = py[k].dump()
        assert np.array_equal(np.array(pt, np.int8), tiles[k]) and pts == int(ts[k])


@pytest.mark.parametrize("seed", range(9))
def test_c_oracle_equals_python_oracle_with_other_player_counts(seed):
    """The same differential check with 1, 3 and 4 players (both oracles are pinned to the reference's N-player fixtures on their
    own, tests/test_oracle_golden.py): the kitchens the GPU test below steps."""
    players = (1, 3, 4)[seed % 3]
    rng = np.random.default_rng(3000 + seed)
    params = random_layout_players(rng, players)
    n, T = 3, 200
    acts = busy_actions_players(rng, T, players, n)
    env = oracle.OracleVecEnv(params, n)
    py = [PyKitchen(params) for _ in range(n)]
    for t in range(T):
        obs, rew, done = env.step(acts[t])
        for k in range(n):
            o, r, d = py[k].n_step([int(acts[t, i, k]) for i in range(players)])
            assert r == int(rew[0, k]) and d == bool(done[k]), (seed, t, k)
            assert np.array_equal(np.array(o, np.int8), obs[:, k]), (seed, t, k)
    tiles, _, ts = env.get_state()
    for k in range(n):
        pt, _, pts = py[k].dump()
        assert np.array_equal(np.array(pt, np.int8), tiles[k]) and pts == int(ts[k])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_hip_equals_oracle_on_random_kitchens(seed, monkeypatch):
    import torch
    from diverse_conventions_amd import OvercookedEngine
    if seed % 2:
        monkeypatch.setenv("OC_NO_JIT", "1")             # alternate between the specialised and the generic build
    rng = np.random.default_rng(2000 + seed)
    params = random_layout(rng)
    n, T = 700 + 13 * seed, 300
    acts = busy_actions(rng, T, n)
    ora = oracle.OracleVecEnv(params, n, nthreads=4)
    eng = OvercookedEngine(params, n, 0)
    dacts = torch.from_numpy(acts).cuda()
    rewards = 0
    for t in range(T):
        obs, rew, done = eng.step(dacts[t])
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), oobs), (seed, t)
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), (seed, t)
        rewards += int(orew[0].sum())
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(9))
def test_hip_equals_oracle_on_random_kitchens_with_other_player_counts(seed):
    """The N-player group kernel (csrc/oc_np_kernels.h) on kitchens nobody shipped: 1, 3 and 4 players, grids up to 10x7 with islands,
    several pots and dispensers, non-uniform recipe tables, horizons of 20-120 steps; odd kitchen counts (ragged last group, viewer
    planes off dword boundaries), every step's observations, rewards and dones against the C oracle, records at the end."""
    import torch
    from diverse_conventions_amd import OvercookedEngine
    players = (1, 3, 4)[seed % 3]
    rng = np.random.default_rng(3000 + seed)
    params = random_layout_players(rng, players)
    n, T = 900 + 37 * seed, 260
    acts = busy_actions_players(rng, T, players, n)
    ora = oracle.OracleVecEnv(params, n, nthreads=4)
    eng = OvercookedEngine(params, n, 0)
    assert np.array_equal(eng.observe().cpu().numpy(), ora.encode())
    dacts = torch.from_numpy(acts).cuda()
    for t in range(T):
        obs, rew, done = eng.step(dacts[t])
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(obs.cpu().numpy(), oobs), (seed, t)
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done.cpu().numpy(), odone), (seed, t)
    for x, y in zip(eng.get_state(), ora.get_state()):
        assert np.array_equal(x, y)
    eng.close()
