"""BASELINE.json configs 4 and 5 at their per-GPU shape, as parity tests (all through the C ABI, on one MI355X):

config 4  asymmetric_advantages, 65 536 kitchens over 8 GPUs -> one GPU's share of 8184 kitchens (31 pairing segments of
          264: k = 15 trained conventions + the one in training, train/XD/serial.py:27), 16-convention paired rollout with
          critics, the mixed-play collection at L = 200 on an engine of the same size (train/XD/xd_player.py:232-356;
          train/XD/serial.py:28 makes the mixed-play env beside the pairing env), and the compact trajectory exchange
          between two shards with receiver-side re-featurisation.
config 5  all five classic layouts, 32 conventions (k = 31 -> 63 segments of 130 kitchens): rollout -> GAE -> one
          minibatch epoch per layout, plus the joint env step of the five engines as one launch.

Everything recorded on the GPU is checked against the CPU oracles: the env oracle driven by the recorded actions, the
reference-architecture PyTorch evaluation of the networks (tolerances written at the assertions), oracle/mixed_oracle.py,
oracle/gae_oracle.py and oracle/minibatch_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import gae_oracle, oracle
from oracle import minibatch_oracle as mo
from oracle.mixed_oracle import buffer_source_steps, policy_choice

pytestmark = pytest.mark.gpu

CLASSIC = ["simple", "unident_s", "random1", "random0", "random3"]


def _random_nets(W, H, n_actors, n_critics, seed):
    P = (W - 2) * (H - 2)
    rng = np.random.default_rng(seed)
    r = lambda *s: (rng.standard_normal(s) * 0.1).astype(np.float32)
    trunk = lambda: {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                     "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64)}
    actors = [dict(trunk(), **{"act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)}) for _ in range(n_actors)]
    critics = [dict(trunk(), **{"v_out.weight": r(1, 64), "v_out.bias": r(1)}) for _ in range(n_critics)]
    return actors, critics


def _check_env_against_oracle(params, traj, T, every=1, start_state=None):
    """The recorded actions replayed through the oracle reproduce every recorded obs / reward / done of every kitchen."""
    n = traj.obs.shape[2]
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    if start_state is not None:
        ora.set_state(*start_state)
    assert np.array_equal(traj.obs[0].cpu().numpy(), ora.encode())
    acts = traj.actions.cpu().numpy()
    rew, done = traj.reward.cpu().numpy(), traj.done.cpu().numpy()
    for t in range(T):
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(rew[t], orew) and np.array_equal(done[t], odone), f"step {t}: reward/done"
        if t % every == 0 or t == T - 1:
            assert np.array_equal(traj.obs[t + 1].cpu().numpy(), oobs), f"step {t}: obs"
    return acts


def _check_policies(ro, traj, logp, values, actors_sd, critics_sd, critic_of, T, shape, lp_tol=8e-6, v_tol=4e-6):
    """Recorded log-probs are those of the recorded actions under the segment's actor, recorded values those of its critic
    (PyTorch f32 evaluation of the reference architecture; the fused kernel agrees to ~1e-6, the tolerance leaves room for
    the different summation order of 21-position grids)."""
    from torch_policies import StackedActors, StackedCritics
    from diverse_conventions_amd.sharding import segment_policy_indices
    W, H, _ = shape
    G, threads, n = len(ro.segments), ro.threads, traj.obs.shape[2]
    sa, sc = StackedActors(actors_sd, shape, "cuda"), StackedCritics(critics_sd, shape, "cuda")
    by_seg = lambda x: x.reshape(T, G, threads, W, H, 20).permute(1, 0, 2, 3, 4, 5).reshape(G, T * threads, W, H, 20)
    for p, side in enumerate(("ego", "partner")):
        obs = by_seg(traj.obs[:T, p])
        sel = sa.select(torch.tensor(segment_policy_indices(ro.segments, side)))
        lp_all = sa.log_probs(obs, selected=sel).view(G, T, threads, 6).permute(1, 0, 2, 3).reshape(T, n, 6)
        got = lp_all.gather(-1, traj.actions[:, p].long().unsqueeze(-1)).squeeze(-1)
        assert (got - logp[:, p]).abs().max() < lp_tol, f"log-probs, side {side}"
        if values is not None:
            csel = sc.select(torch.tensor([c - len(actors_sd) for c in critic_of[p]]))
            v = sc.values(obs, selected=csel).view(G, T, threads).permute(1, 0, 2).reshape(T, n)
            assert (v - values[:, p]).abs().max() < v_tol, f"values, side {side}"


def test_config4_one_gpu_share(layout_params):
    from diverse_conventions_amd import (FusedPairedRollout, MixedPlayRollout, OvercookedEngine, PackedNetworks, TrajectoryBuffer,
                                         TrajectoryExchange)
    from torch_policies import StackedActors, StackedCritics
    params = layout_params["unident_s"]                       # asymmetric_advantages
    W, H, shape = 9, 5, (9, 5, 20)
    k, G = 15, 31
    threads = 8192 // G                                       # 264
    n = threads * G                                           # 8184 of the GPU's 8192 kitchens
    actors_sd, critics_sd = _random_nets(W, H, k + 1, G + 1, seed=44)        # 16 conventions; a critic per segment + the mp critic
    nets = PackedNetworks.from_state_dicts(shape, "cuda", actors=actors_sd, critics=critics_sd)
    critic_of = [[k + 1 + g for g in range(G)]] * 2

    # ---- (a) the paired rollout: 16 conventions x both sides x 31 segments in one policy launch + one env launch per step
    T = 24
    eng = OvercookedEngine(params, n, 0)
    ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=critic_of, seed=5)
    assert len(ro.segments) == G
    traj = TrajectoryBuffer(eng, T, store_states=True)        # compact records written by the env launches themselves (oc_step's d_records)
    logp = torch.zeros((T, 2, n), device="cuda")
    values = torch.zeros((T, 2, n), device="cuda")
    ro.collect(traj, logp=logp, values=values)
    acts = _check_env_against_oracle(params, traj, T)
    assert len(np.unique(acts)) == 6
    _check_policies(ro, traj, logp, values, actors_sd, critics_sd, critic_of, T, shape)
    # the recorded trajectory IS the wire buffer: one flat int32 allocation in the exchange's order, every slot the state its
    # observation slot shows, the last one the engine's current state
    assert traj.records.data_ptr() == traj.wire.data_ptr() and traj.done.data_ptr() + traj.done.numel() * 4 <= traj.wire.data_ptr() + traj.wire.numel() * 4
    assert torch.equal(traj.records[T], eng.export_states())
    ex1 = TrajectoryExchange()
    got1 = ex1.gather_buffer(traj)                            # world 1: views of the wire buffer itself, no copy
    assert got1["records"].shape == (1, T + 1, n, 52) and got1["records"].data_ptr() == traj.wire.data_ptr()
    assert torch.equal(ex1.featurize_all(eng, got1["records"]), traj.obs)
    assert torch.equal(got1["actions"][0], traj.actions) and torch.equal(got1["reward"][0], traj.reward) and torch.equal(got1["done"][0], traj.done)

    # ---- (b) the compact exchange of that trajectory between two shards, receivers re-featurise
    world, half = 2, n // 2
    shards = [OvercookedEngine(params, half, 0) for _ in range(world)]
    dacts = traj.actions
    packs = []
    for r, sh in enumerate(shards):
        rec = torch.zeros((T, half, sh.record_bytes // 4), dtype=torch.int32, device="cuda")
        rw = torch.zeros((T, half), dtype=torch.int32, device="cuda")
        dn = torch.zeros((T, half), dtype=torch.int32, device="cuda")
        a_sh = dacts[:, :, r * half:(r + 1) * half].contiguous()
        for t in range(T):
            sh.export_states(out=rec[t])                      # the state whose observation is obs[t]
            _, rew_t, done_t = sh.step(a_sh[t])
            rw[t].copy_(rew_t[0]); dn[t].copy_(done_t)
        ex = TrajectoryExchange()
        got = ex.gather_trajectory(rec, a_sh, rw, dn)         # world 1 here: [1, ...] views of one packed buffer
        assert got["records"].shape == (1, T, half, 52) and got["records"].data_ptr() % 16 == 0
        packs.append(got)
    assert eng.record_bytes == 208                            # 224 B per kitchen-step on the wire: 367 MB at T=200 x 8192 (334 MB)
    gathered = torch.cat([p["records"] for p in packs])       # what all_gather_into_tensor yields: [world, T, n/2, rec/4]
    all_obs = TrajectoryExchange().featurize_all(shards[0], gathered)
    assert all_obs.shape == (T, 2, n, W, H, 20)
    assert torch.equal(all_obs, traj.obs[:T])
    assert torch.equal(torch.cat([p["reward"][0] for p in packs], dim=1), traj.reward[:, 0])
    assert torch.equal(torch.cat([p["done"][0] for p in packs], dim=1), traj.done)
    for sh in shards:
        sh.close()
    eng.close()
    del traj, logp, values, all_obs, gathered

    # ---- (c) mixed play at L = 200 on an engine of the same size: 2L steps, per-row policy mixing, staircase buffer
    L = 200
    eng = OvercookedEngine(params, n, 0)
    best_i, mp_critic = 7, k + 1 + G                          # the partner convention and the mixed-play critic
    mp = MixedPlayRollout(eng, nets, L, main=(0, mp_critic), partner=(1 + best_i, k + 1 + 3), deterministic=True)
    g = torch.Generator(device="cuda").manual_seed(11)
    uniforms = torch.rand((2 * L, 2, n), device="cuda", generator=g)
    mp.collect(uniforms)
    reps = -(-n // (L - 1))
    ora = oracle.OracleVecEnv(params, n, nthreads=8)
    assert np.array_equal(mp.traj.obs[0].cpu().numpy(), ora.encode())
    acts = mp.traj.actions.cpu().numpy()
    rew, done = mp.traj.reward.cpu().numpy(), mp.traj.done.cpu().numpy()
    sa = StackedActors([actors_sd[0], actors_sd[1 + best_i]], shape, "cuda")
    sc = StackedCritics([critics_sd[mp_critic - k - 1], critics_sd[3]], shape, "cuda")
    u_host = uniforms.cpu().numpy()
    choice = mp.choice.cpu().numpy()
    for i in range(2 * L):
        for side in range(2):
            # the reference's rule, one environment per stretch of `reps` kitchens (every reps-th kitchen = one full copy of its envs)
            for off in (0, reps // 2):
                mine = u_host[i, side, off::reps]                         # one kitchen per environment the batch covers
                draws = np.ones(L - 1, dtype=np.float32)
                draws[:len(mine)] = mine
                assert np.array_equal(choice[i, side, off::reps], policy_choice(L, i, draws)[:len(mine)]), (i, side)
        if i % 25 == 0 or i in (L - 1, L, 2 * L - 1):         # per-network PyTorch evaluation of the step's observations
            for side in range(2):
                obs = mp.traj.obs[i, side].unsqueeze(0)
                a = torch.stack([sa.act(obs, torch.tensor([q]), deterministic=True)[0].view(-1) for q in range(2)])
                v = torch.stack([sc.values(obs, torch.tensor([q])).view(-1) for q in range(2)])
                sel = mp.choice[i, side].long().unsqueeze(0)
                assert np.array_equal(a.gather(0, sel)[0].cpu().numpy(), acts[i, side]), (i, side)
                assert (v.gather(0, sel)[0] - mp.values[i, side]).abs().max() < 4e-6
        oobs, orew, odone = ora.step(acts[i])
        assert np.array_equal(rew[i], orew) and np.array_equal(done[i], odone), i
        assert int(odone.sum()) == (n if i in (L - 1, 2 * L - 1) else 0)
        if i % 10 == 0 or i >= 2 * L - 2:
            assert np.array_equal(mp.traj.obs[i + 1].cpu().numpy(), oobs), i
    buf = mp.buffer()
    src = np.repeat(buffer_source_steps(L), reps, axis=1)[:, :n]          # kitchen k plays environment k // reps
    assert np.array_equal(buf["source_step"].cpu().numpy(), src)
    for t in (0, 1, 57, L - 1):
        w = torch.from_numpy(src[t]).cuda()
        kk = torch.arange(n, device="cuda")
        for p in range(2):
            assert torch.equal(buf["obs"][t, p], mp.traj.obs[w, p, kk]) and torch.equal(buf["actions"][t, p], mp.traj.actions[w, p, kk])
            assert torch.equal(buf["value_preds"][t, p], mp.values[w, p, kk])
    assert int(buf["choice"].abs().sum()) == 0                            # every kept step was played by the trained policy
    eng.close()


def test_config5_shape_rollout_gae_minibatch(layout_params):
    """Five layouts x 32 conventions: per layout 63 pairing segments of 130 kitchens (8190 of 8192), rollout with policies
    and critics in the loop, GAE, advantage normalisation and one feed-forward minibatch epoch -- each stage against its
    oracle -- then the five engines' joint env step as ONE launch against the env oracle."""
    from diverse_conventions_amd import (EngineGroup, FusedPairedRollout, MinibatchSource, OvercookedEngine, PackedNetworks,
                                         TrajectoryBuffer, gae_returns, normalize_advantages)
    k, G, T = 31, 63, 12
    threads = 8192 // G                                       # 130
    n = threads * G                                           # 8190
    engines, oras, kept = [], [], []
    for li, name in enumerate(CLASSIC):
        params = layout_params[name]
        W, H = params["width"], params["height"]
        shape = (W, H, 20)
        actors_sd, critics_sd = _random_nets(W, H, k + 1, 4, seed=100 + li)
        nets = PackedNetworks.from_state_dicts(shape, "cuda", actors=actors_sd, critics=critics_sd)
        critic_of = [[k + 1 + (g % 4) for g in range(G)], [k + 1 + ((g + 1) % 4) for g in range(G)]]
        eng = OvercookedEngine(params, n, 0)
        ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=critic_of, seed=li)
        assert len(ro.segments) == G and max(s.ego_policy for s in ro.segments) == k
        traj = TrajectoryBuffer(eng, T)
        logp = torch.zeros((T, 2, n), device="cuda")
        values = torch.zeros((T + 1, 2, n), device="cuda")
        ro.collect(traj, logp=logp, values=values[:T])
        _check_env_against_oracle(params, traj, T, every=4)
        _check_policies(ro, traj, logp, values[:T], actors_sd, critics_sd, critic_of, T, shape)
        # ---- GAE over the collected trajectory (reference layout [T, N, A, 1]); bootstrap value = a last critic pass
        ro.plan.forward(traj.obs[T], raw=values[T].view(-1), actions=torch.zeros(2 * n, dtype=torch.int32, device="cuda"),
                        deterministic=True)
        ref = lambda x: x.permute(0, 2, 1).unsqueeze(-1).contiguous()
        rewards = ref(traj.reward.float())
        vals = ref(values)
        masks = torch.ones((T + 1, n, 2, 1), device="cuda")
        masks[1:] = ref((1 - traj.done).float().unsqueeze(1).expand(-1, 2, -1))
        ret, adv = gae_returns(rewards, vals, masks, value_normalizer=(0.1, 1.3))
        cols = np.random.default_rng(li).choice(n, size=300, replace=False)
        r_o, a_o = gae_oracle.gae_returns(rewards[:, cols].cpu().numpy(), vals[:, cols].cpu().numpy(), masks[:, cols].cpu().numpy(),
                                          None, 0.99, 0.95, True, 0.1, 1.3)
        assert np.array_equal(ret[:-1, cols].cpu().numpy(), r_o) and np.array_equal(adv[:, cols].cpu().numpy(), a_o)
        # ---- normalisation + one minibatch epoch, against the numpy restatement of the reference's generator
        norm, stats = normalize_advantages(adv)
        want_norm, mean, std = mo.normalize_advantages(adv.cpu().numpy())
        assert abs(float(stats[0]) - mean) < 2e-6 and abs(float(stats[1]) - std) < 2e-6          # f64 sums both sides
        np.testing.assert_allclose(norm.cpu().numpy(), want_norm, rtol=0, atol=1e-5)
        src = MinibatchSource.from_trajectory(traj, vals, ret, masks, ref(logp))
        perm = torch.randperm(T * n * 2, generator=torch.Generator().manual_seed(li))
        buf = {"obs": traj.obs.permute(0, 2, 1, 3, 4, 5).cpu().numpy(), "actions": src.cols["actions"].cpu().numpy(),
               "value_preds": vals.cpu().numpy(), "returns": ret.cpu().numpy(), "masks": masks.cpu().numpy(),
               "active_masks": np.ones((T + 1, n, 2, 1), np.float32), "action_log_probs": ref(logp).cpu().numpy()}
        got = list(src.feed_forward_generator(norm, 4, perm=perm, obs_dtype=torch.int8))
        want = list(mo.feed_forward_batches(buf, norm.cpu().numpy(), perm.numpy(), 4))
        assert len(got) == 4
        pos = dict(obs=1, actions=4, value_preds=5, returns=6, masks=7, active_masks=8, action_log_probs=9, adv_targ=10)
        for b, w in zip(got, want):
            for f, i in pos.items():
                assert np.array_equal(b[i].cpu().numpy(), w[f]), (name, f)
        engines.append(eng)
        kept.append((nets, actors_sd, critics_sd, critic_of, shape))
        oras.append((params, traj.actions[T - 1].clone()))
        del traj, logp, values, src, buf, got, want
    # ---- the joint env step of the five engines at this shape: one launch (n * S * 20 is not a multiple of 16 for the 5x5
    # grids, so the second viewer's plane starts 8 bytes off a 16-byte boundary)
    grp = EngineGroup(engines)
    rng = np.random.default_rng(77)
    for step in range(3):
        acts = [torch.from_numpy(rng.integers(0, 6, size=(2, n)).astype(np.int32)).cuda() for _ in engines]
        before = [e.get_state() for e in engines]
        obs, rew, done = grp.step(acts)
        for e, a, st, (params, _), o, r, d in zip(engines, acts, before, oras, obs, rew, done):
            ora = oracle.OracleVecEnv(params, n, nthreads=8)
            ora.set_state(*st)
            oobs, orew, odone = ora.step(a.cpu().numpy())
            assert np.array_equal(o.cpu().numpy(), oobs) and np.array_equal(r.cpu().numpy(), orew) and np.array_equal(d.cpu().numpy(), odone)
    # ---- and the whole five-layout rollout with the policies in the loop through the group API (every engine's policy + env
    # launches on its own stream, enqueued from C): continues from the engines' current states
    from diverse_conventions_amd import GroupPairedRollout
    T2 = 5
    gro = GroupPairedRollout(grp, [kp[0] for kp in kept], threads, k, critic_nets=[kp[3] for kp in kept], seed=3)
    starts = [e.get_state() for e in engines]
    trajs = [TrajectoryBuffer(e, T2) for e in engines]
    lps = [torch.zeros((T2, 2, n), device="cuda") for _ in engines]
    vas = [torch.zeros((T2, 2, n), device="cuda") for _ in engines]
    gro.collect(trajs, logps=lps, values=vas)
    for i, (e, (params, _)) in enumerate(zip(engines, oras)):
        _check_env_against_oracle(params, trajs[i], T2, start_state=starts[i])
        shim = type("Shim", (), {"segments": gro.segments[i], "threads": threads})()
        _check_policies(shim, trajs[i], lps[i], vas[i], kept[i][1], kept[i][2], kept[i][3], T2, kept[i][4])
    gro.close()
    grp.close(close_engines=True)


def test_policy_rollout_into_fresh_slots_at_the_large_batch_geometry(layout_params):
    """A policy rollout that COLLECTS (every step's planes into their own time-major slot) on a batch large enough for the
    fresh-destination module (16 kitchens per group, eight wavefronts per CU): while the rollout runs the planes are stored plainly --
    the next launch, the policy kernel, reads them (`_pl` entry points, oc_internal_planes_read_next) -- instead of streamed past the
    caches.  3 x 64 sampled kitchens of the recorded trajectory replayed through the env oracle, across an episode boundary; the
    recorded log-probabilities against the per-segment PyTorch evaluation on the sampled rows."""
    from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, TrajectoryBuffer
    params = dict(layout_params["simple"], horizon=6)
    k, threads, T = 2, 10240, 9                                       # 5 segments x 10 240 = 51 200 kitchens (>= 49 137: the fresh geometry)
    G, n = 2 * k + 1, threads * (2 * k + 1)
    actors, _ = _random_nets(5, 4, k + 1, 0, 77)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=actors)
    eng = OvercookedEngine(params, n, 0)
    assert "specialised" in eng.kernel_build_info
    ro = FusedPairedRollout(eng, nets, threads, k, seed=3)
    ro.step(2)                                                        # (in place first: the collection then starts from a used engine)
    firsts = [0, n // 2 - 32, n - 64]
    starts = [eng.get_state(f, 64) for f in firsts]
    traj = TrajectoryBuffer(eng, T)
    logp = torch.zeros((T, 2, n), device="cuda")
    ro.collect(traj, logp=logp)
    torch.cuda.synchronize()
    for f, (tiles, players, ts) in zip(firsts, starts):
        ora = oracle.OracleVecEnv(params, 64)
        ora.set_state(tiles, players, ts)
        sl = slice(f, f + 64)
        assert np.array_equal(traj.obs[0][:, sl].cpu().numpy(), ora.encode())
        for t in range(T):
            oobs, orew, odone = ora.step(np.ascontiguousarray(traj.actions[t][:, sl].cpu().numpy()))
            assert np.array_equal(traj.obs[t + 1][:, sl].cpu().numpy(), oobs), (f, t)
            assert np.array_equal(traj.reward[t][:, sl].cpu().numpy(), orew) and np.array_equal(traj.done[t][sl].cpu().numpy(), odone), (f, t)
    assert int(traj.done.sum()) > 0                                   # the horizon was crossed inside the collection
    # the log-probabilities recorded for the sampled kitchens' actions: the segment's actor on the recorded observations
    from torch_policies import StackedActors
    from diverse_conventions_amd.sharding import segment_policy_indices
    sa = StackedActors(actors, (5, 4, 20), "cuda")
    for p, side in enumerate(("ego", "partner")):
        pol = segment_policy_indices(ro.segments, side)
        for f in firsts:
            g = f // threads
            assert (f + 63) // threads == g
            obs = traj.obs[:T, p, f:f + 64].reshape(1, T * 64, 5, 4, 20)
            lp = sa.log_probs(obs, selected=sa.select(torch.tensor([pol[g]]))).view(T, 64, 6)
            got = lp.gather(-1, traj.actions[:, p, f:f + 64].long().unsqueeze(-1)).squeeze(-1)
            assert (got - logp[:, p, f:f + 64]).abs().max() < 8e-6
    eng.close()


def test_collecting_into_a_single_slot_trajectory_is_refused(layout_params):
    """`TrajectoryBuffer(obs_slots=1)` holds ONE observation slot; the native policy rollouts write slot t+1 at step t, T slots past
    its end.  Every collecting entry point refuses such a buffer (and one that belongs to another engine) before anything is enqueued."""
    from diverse_conventions_amd import (EngineGroup, FusedPairedRollout, GroupPairedRollout, OvercookedEngine, PackedNetworks,
                                         TrajectoryBuffer)
    params = layout_params["simple"]
    k, threads, T = 1, 40, 5
    n = threads * (2 * k + 1)
    actors, _ = _random_nets(5, 4, k + 1, 0, 5)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=actors)
    eng, other = OvercookedEngine(params, n, 0), OvercookedEngine(params, n, 0)
    ro = FusedPairedRollout(eng, nets, threads, k, seed=1)
    one = TrajectoryBuffer(eng, T, obs_slots=1)
    canary = one.obs.clone()
    with pytest.raises(ValueError, match="obs_slots"):
        ro.collect(one)
    with pytest.raises(ValueError, match="obs_slots"):
        one.rollout(torch.zeros((T, 2, n), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError, match="another engine"):
        ro.collect(TrajectoryBuffer(other, T))
    grp = EngineGroup([eng])
    gro = GroupPairedRollout(grp, [nets], [threads], k, seed=1)
    with pytest.raises(ValueError, match="obs_slots"):
        gro.collect([one])
    with pytest.raises(ValueError, match="one trajectory per engine"):
        gro.collect([TrajectoryBuffer(other, T)])
    torch.cuda.synchronize()
    assert torch.equal(one.obs, canary)
    ro.collect(TrajectoryBuffer(eng, T))                      # (the ordinary buffer still collects)
    grp.close()
    eng.close(); other.close()
