"""GPU checks of the pieces that sit on either side of the rollout: peers' shards re-featurised from
compact records (the receive side of the RCCL all-gather), and BASELINE config 3's shape -- the five
classic layouts x 8192 kitchens with the reference's convention-pairing segments."""
import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu


def test_peer_shards_refeaturised_in_place(layout_params):
    from diverse_conventions_amd import KitchenShard, OvercookedEngine, TrajectoryExchange
    params = layout_params["random3"]
    total, world, T = 96, 2, 25
    rng = np.random.default_rng(2)
    acts = rng.integers(0, 6, size=(T, 2, total)).astype(np.int32)
    ref = oracle.OracleVecEnv(params, total)
    engines = [OvercookedEngine(params, KitchenShard(total, world, r).count, 0) for r in range(world)]
    recs = [[] for _ in range(world)]
    want = []
    for t in range(T):
        for r, eng in enumerate(engines):
            sh = KitchenShard(total, world, r)
            eng.step(torch.from_numpy(acts[t][:, sh.first:sh.first + sh.count].copy()).cuda())
            recs[r].append(eng.export_states())
        want.append(ref.step(acts[t])[0].copy())
    # what all_gather_into_tensor yields on every rank: [world, T, n, rec/4]
    gathered = torch.stack([torch.stack(r) for r in recs])
    ex = TrajectoryExchange()
    assert ex.world == 1 and ex.all_gather(recs[0][0]).shape[0] == 1
    for t in (0, 7, T - 1):
        obs = ex.featurize(engines[0], gathered, t)
        assert obs.shape == (2, total, 8, 5, 20)
        assert np.array_equal(obs.cpu().numpy(), want[t]), t
    # the whole trajectory of every peer: one launch per peer for all T steps
    all_obs = ex.featurize_all(engines[0], gathered)
    assert all_obs.shape == (T, 2, total, 8, 5, 20)
    assert np.array_equal(all_obs.cpu().numpy(), np.stack(want))
    for e in engines:
        e.close()


def test_config3_five_layouts_with_pairing_segments(layout_params):
    """5 classic layouts x 8192 kitchens, 8 conventions -> 15 segments of 546 (+2 spare) kitchens.
    Each (policy, side) pair draws its own action stream, so segments really differ; a sample of
    kitchens per layout is checked against the oracle, and every engine steps on its own stream."""
    from diverse_conventions_amd import CLASSIC_LAYOUTS, OvercookedEngine, pairing_segments
    n, k, T = 8192, 7, 210
    threads = n // (2 * k + 1)
    segs = pairing_segments(threads, k)
    assert len(segs) == 15 and segs[-1].stop <= n
    rng = np.random.default_rng(9)
    # one action tape per policy (index 0 = learner, 1..7 frozen conventions) and side
    tapes = rng.integers(0, 6, size=(k + 1, 2, T, threads)).astype(np.int32)
    acts = np.full((T, 2, n), 4, dtype=np.int32)
    for s in segs:
        acts[:, 0, s.start:s.stop] = tapes[s.ego_policy, 0]
        acts[:, 1, s.start:s.stop] = tapes[s.partner_policy, 1]
    sample = np.sort(rng.choice(n, size=256, replace=False))
    engines, oracles, streams = [], [], []
    for name in CLASSIC_LAYOUTS:
        engines.append(OvercookedEngine(layout_params[name], n, 0))
        oracles.append(oracle.OracleVecEnv(layout_params[name], len(sample), nthreads=4))
        streams.append(torch.cuda.Stream())
    dacts = torch.from_numpy(acts).cuda()
    idx = torch.from_numpy(sample).cuda()
    torch.cuda.synchronize()
    for t in range(T):
        for eng, st in zip(engines, streams):
            with torch.cuda.stream(st):
                eng.step(dacts[t])
        torch.cuda.synchronize()
        for eng, ora in zip(engines, oracles):
            oobs, orew, odone = ora.step(acts[t][:, sample])
            assert np.array_equal(eng.obs[:, idx].cpu().numpy(), oobs), t
            assert np.array_equal(eng.reward[:, idx].cpu().numpy(), orew) and np.array_equal(eng.done[idx].cpu().numpy(), odone)
    # SP kitchens of one segment got identical tapes on both sides only through policy 0
    for eng in engines:
        assert "specialised" in eng.kernel_build_info or "OC_NO_JIT" in eng.kernel_build_info or "OC_ABLATE" in eng.kernel_build_info
        eng.close()


def test_trajectory_buffer_is_filled_in_place(layout_params):
    """f1: the kernel's destination is the trajectory slot itself (obs int8 [T+1,P,N,W,H,C]) -- stepwise with
    recorded states, and as one C-side rollout; both equal the oracle, and the records re-encode to the obs."""
    from diverse_conventions_amd import OvercookedEngine, TrajectoryBuffer
    params = layout_params["simple"]
    n, T = 333, 205
    rng = np.random.default_rng(12)
    acts = rng.integers(0, 6, size=(T, 2, n)).astype(np.int32)
    ora = oracle.OracleVecEnv(params, n)
    want = [ora.encode().copy()]
    rews, dones = [], []
    for t in range(T):
        o, r, d = ora.step(acts[t])
        want.append(o.copy()); rews.append(r.copy()); dones.append(d.copy())
    eng = OvercookedEngine(params, n, 0)
    buf = TrajectoryBuffer(eng, T, store_states=True)
    dacts = torch.from_numpy(acts).cuda()
    for t in range(T):
        o, r, d = buf.step(dacts[t].to(torch.int64).unsqueeze(-1))
        assert o.data_ptr() == buf.obs[t + 1].data_ptr()          # a view of the slot, not a copy
    assert np.array_equal(buf.obs.cpu().numpy(), np.stack(want))
    assert np.array_equal(buf.reward.cpu().numpy(), np.stack(rews)) and np.array_equal(buf.done.cpu().numpy(), np.stack(dones))
    assert np.array_equal(buf.actions.cpu().numpy(), acts)
    for t in (0, 77, T):
        assert np.array_equal(eng.encode_states(buf.records[t]).cpu().numpy(), want[t])
    # the records were written by the step launches themselves (oc_step's d_records): every slot equals the oracle's state after
    # that many steps, and the compact fields are views of ONE flat buffer in the exchange's wire order
    from diverse_conventions_amd.engine import pack_states
    chk = oracle.OracleVecEnv(params, n)
    recs = buf.records.cpu().numpy()
    for t in range(T + 1):
        if t:
            chk.step(acts[t - 1])
        assert np.array_equal(recs[t].view(np.uint32), pack_states(*chk.get_state(), eng.record_bytes)), t
    assert buf.records.data_ptr() == buf.wire.data_ptr() and buf.wire.numel() * 4 == (T + 1) * n * eng.record_bytes + (2 + 2 + 1) * T * n * 4 + 4 * sum(
        (-w) % 4 for w in (T * 2 * n, T * 2 * n, T * n))
    assert torch.equal(buf.records[T], eng.export_states())
    wire_stepwise = buf.wire.clone()                          # (after_update below moves the last slot to the front)
    with pytest.raises(IndexError):
        buf.step(dacts[0])
    buf.after_update()
    assert np.array_equal(buf.obs[0].cpu().numpy(), want[T]) and buf.t == 0
    eng2 = OvercookedEngine(params, n, 0)
    buf2 = TrajectoryBuffer(eng2, T).rollout(dacts)
    assert np.array_equal(buf2.obs.cpu().numpy(), np.stack(want)) and np.array_equal(buf2.reward.cpu().numpy(), np.stack(rews))
    assert buf2.agent_view(1).shape == (T + 1, n, 5, 4, 20)
    # the same trajectory, records included, as T launches enqueued by one C call (oc_rollout), as ONE launch (oc_rollout_fused: the
    # records leave the LDS-resident state once per step), and with a single observation slot overwritten in place
    for how in ("tape", "fused", "one_slot"):
        e3 = OvercookedEngine(params, n, 0)
        b3 = TrajectoryBuffer(e3, T, store_states=True, obs_slots=1 if how == "one_slot" else None)
        b3.rollout(dacts) if how == "fused" else b3.rollout_tape(dacts)
        assert torch.equal(b3.wire, wire_stepwise), how
        if how == "one_slot":
            assert b3.obs.shape[0] == 1 and np.array_equal(b3.obs[0].cpu().numpy(), want[T])
        else:
            assert np.array_equal(b3.obs.cpu().numpy(), np.stack(want)), how
        e3.close()
    eng.close(); eng2.close()


def test_paired_rollout_eager_and_graph(layout_params):
    """f2: batched policies for both sides + env step; deterministic actors so that eager, HIP-graph replay and
    an oracle driven by the same (separately evaluated) actions agree exactly."""
    import os
    from conftest import GOLDEN
    from diverse_conventions_amd import OvercookedEngine
    from torch_policies import StackedActors
    from torch_policies import PairedRollout
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    sds = [{k[len(f"a{i}."):]: z[k] for k in z.files if k.startswith(f"a{i}.") and "logits" not in k and "det_" not in k}
           for i in range(3)]
    params = layout_params["simple"]
    threads, k, T = 64, 2, 40
    n = threads * (2 * k + 1)
    actors = StackedActors(sds, (5, 4, 20), "cuda")
    runs = []
    for use_graph in (False, True):
        eng = OvercookedEngine(params, n, 0)
        ro = PairedRollout(eng, actors, threads, k, deterministic=True)
        if use_graph:
            ro.capture()
            eng.reset_all()                      # capture warm-up stepped the kitchens
        ora = oracle.OracleVecEnv(params, n)
        for t in range(T):
            prev = eng.obs.clone()
            obs, rew, done = ro.step()
            acts = ro.actions.cpu().numpy()
            # the same actions from per-segment, per-policy evaluation (what the reference does sequentially)
            for p in range(2):
                for g, seg in enumerate(ro.segments):
                    pol = seg.ego_policy if p == 0 else seg.partner_policy
                    a, _ = actors.act(prev[p, seg.start:seg.stop].unsqueeze(0), torch.tensor([pol]), deterministic=True)
                    assert np.array_equal(a.view(-1).cpu().numpy(), acts[p, seg.start:seg.stop]), (t, p, g)
            oobs, orew, odone = ora.step(acts)
            assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew), t
        runs.append(eng.obs.cpu().numpy().copy())
        eng.close()
    assert np.array_equal(runs[0], runs[1])


def _golden_nets():
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    pick = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and ("weight" in k or "bias" in k)}
    return [pick(f"a{i}.") for i in range(3)], [pick(f"c{i}.") for i in range(2)]


def test_fused_paired_rollout_eager_graph_and_reference_policies(layout_params):
    """f2 on the MFMA kernel: one policy launch + one env launch per step.  Deterministic actors, so the fused
    rollout, its HIP-graph replay, the batched-GEMM PyTorch path and an oracle env driven by the same actions
    must agree exactly; critic values against the reference-architecture PyTorch evaluation within 2e-6."""
    from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks
    from torch_policies import StackedActors, StackedCritics
    actors_sd, critics_sd = _golden_nets()
    params = layout_params["simple"]
    threads, k, T = 64, 2, 40
    n = threads * (2 * k + 1)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=actors_sd, critics=critics_sd)
    sa, sc = StackedActors(actors_sd, (5, 4, 20), "cuda"), StackedCritics(critics_sd, (5, 4, 20), "cuda")
    critic_of = [[3, 4, 3, 4, 3], [4, 4, 3, 3, 4]]                   # arbitrary critic per (side, segment); nets 3, 4
    runs = []
    for use_graph in (False, True):
        eng = OvercookedEngine(params, n, 0)
        ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=critic_of, deterministic=True, rows_per_tile=32)
        if use_graph:
            ro.capture()
            eng.reset_all()
        ora = oracle.OracleVecEnv(params, n)
        for t in range(T):
            prev = eng.obs.clone()
            obs, rew, done = ro.step()
            acts = ro.actions.cpu().numpy()
            for p in range(2):
                for g, seg in enumerate(ro.segments):
                    pol = seg.ego_policy if p == 0 else seg.partner_policy
                    o = prev[p, seg.start:seg.stop].unsqueeze(0)
                    a, lp = sa.act(o, torch.tensor([pol]), deterministic=True)
                    assert np.array_equal(a.view(-1).cpu().numpy(), acts[p, seg.start:seg.stop]), (t, p, g)
                    assert (lp.view(-1) - ro.logp[p, seg.start:seg.stop]).abs().max() < 4e-6
                    v = sc.values(o, torch.tensor([critic_of[p][g] - 3]))
                    assert (v.view(-1) - ro.values[p, seg.start:seg.stop]).abs().max() < 2e-6
            oobs, orew, odone = ora.step(acts)
            assert np.array_equal(obs.cpu().numpy(), oobs) and np.array_equal(rew.cpu().numpy(), orew), t
        runs.append(eng.obs.cpu().numpy().copy())
        eng.close()
    assert np.array_equal(runs[0], runs[1])


def test_fused_rollout_collects_a_trajectory_in_place(layout_params):
    """collect(): policies read trajectory slot t and write actions into the trajectory, the env step writes slot
    t+1 -- the recorded actions replayed through the oracle reproduce every recorded observation/reward/done, the
    recorded log-probs are those of the recorded actions, and sampled graph replays differ from step to step."""
    from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, TrajectoryBuffer
    from torch_policies import StackedActors
    actors_sd, critics_sd = _golden_nets()
    params = layout_params["simple"]
    threads, k, T = 96, 1, 30
    n = threads * (2 * k + 1)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=actors_sd[:2], critics=critics_sd[:1])
    eng = OvercookedEngine(params, n, 0)
    ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=[[2] * 3, [2] * 3], seed=11)
    traj = TrajectoryBuffer(eng, T)
    logp = torch.zeros((T, 2, n), device="cuda")
    values = torch.zeros((T, 2, n), device="cuda")
    ro.collect(traj, logp=logp, values=values)
    ora = oracle.OracleVecEnv(params, n)
    assert np.array_equal(traj.obs[0].cpu().numpy(), ora.encode())
    acts = traj.actions.cpu().numpy()
    for t in range(T):
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(traj.obs[t + 1].cpu().numpy(), oobs) and np.array_equal(traj.reward[t].cpu().numpy(), orew)
        assert np.array_equal(traj.done[t].cpu().numpy(), odone)
    sa = StackedActors(actors_sd[:2], (5, 4, 20), "cuda")
    for p in range(2):
        for g, seg in enumerate(ro.segments):
            pol = seg.ego_policy if p == 0 else seg.partner_policy
            o = traj.obs[:T, p, seg.start:seg.stop].reshape(1, T * threads, 5, 4, 20)
            lp_all = sa.log_probs(o, torch.tensor([pol])).view(T, threads, 6)
            a = traj.actions[:, p, seg.start:seg.stop].long()
            assert (lp_all.gather(-1, a.unsqueeze(-1)).squeeze(-1) - logp[:, p, seg.start:seg.stop]).abs().max() < 4e-6
    assert len(np.unique(acts)) == 6 and (acts[1:] != acts[:-1]).mean() > 0.3          # really sampled, step to step
    mean_ret, count = traj.episode_returns()                       # horizon 200 > T: no episode ended inside the buffer
    assert int(count.sum()) == 0 and bool(torch.isnan(mean_ret).all())
    assert ro.steps_done == T
    eng.close()


@pytest.mark.parametrize("layout", ["random0", "random3", "unident_s", "scenario3", "small_corridor", "corridor"])
def test_fused_rollout_on_every_lds_plan(layout, layout_params):
    """The policy kernel chooses a different LDS / wavefront plan per grid size (8 wavefronts with conv digits in LDS;
    4 with them in LDS; 4 or 2 with them in registers; 1 for 14x9): each one in the integrated rollout, sampled actions
    re-evaluated by the PyTorch path on the recorded observations, env against the oracle."""
    from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, PackedNetworks, TrajectoryBuffer
    from torch_policies import StackedActors
    from diverse_conventions_amd.sharding import segment_policy_indices
    from diverse_conventions_amd import get_base_layout_params
    params = layout_params[layout] if layout in layout_params else get_base_layout_params(layout, 200)
    W, H = params["width"], params["height"]
    P = (W - 2) * (H - 2)
    rng = np.random.default_rng(W * 31 + H)
    r = lambda *s: (rng.standard_normal(s) * 0.1).astype(np.float32)
    sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
            "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
            "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(2)]
    threads, k, T = 70, 1, 12
    n = threads * 3
    eng = OvercookedEngine(params, n, 0)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
    ro = FusedPairedRollout(eng, nets, threads, k, seed=4)
    traj = TrajectoryBuffer(eng, T)
    logp = torch.zeros((T, 2, n), device="cuda")
    ro.collect(traj, logp=logp)
    ora = oracle.OracleVecEnv(params, n)
    assert np.array_equal(traj.obs[0].cpu().numpy(), ora.encode())
    acts = traj.actions.cpu().numpy()
    for t in range(T):
        oobs, orew, odone = ora.step(acts[t])
        assert np.array_equal(traj.obs[t + 1].cpu().numpy(), oobs) and np.array_equal(traj.reward[t].cpu().numpy(), orew), t
    sa = StackedActors(sds, (W, H, 20), "cuda")
    for p, side in enumerate(("ego", "partner")):
        sel = sa.select(torch.tensor(segment_policy_indices(ro.segments, side)))
        lp_all = sa.log_probs(traj.obs[:T, p].reshape(T, 3, threads, W, H, 20).permute(1, 0, 2, 3, 4, 5).reshape(3, T * threads, W, H, 20),
                              selected=sel).view(3, T, threads, 6).permute(1, 0, 2, 3).reshape(T, n, 6)
        got = lp_all.gather(-1, traj.actions[:, p].long().unsqueeze(-1)).squeeze(-1)
        assert (got - logp[:, p]).abs().max() < 8e-6
    assert len(np.unique(acts)) == 6
    eng.close()
