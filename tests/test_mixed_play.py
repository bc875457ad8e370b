"""Mixed-play bookkeeping: the reference's `mix_mask` rule and `diaginsert` / `partinsert` placement as
device-side index arithmetic.  tests/golden/mixed_play_cases.npz was recorded from the reference's own XDPlayer /
MixedAgent / SharedReplayBuffer classes (tests/golden/make_golden_mixed.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from diverse_conventions_amd.mixed_play import MixedPlaySchedule
from oracle.mixed_oracle import buffer_source_steps, policy_choice


def golden():
    z = np.load(os.path.join(GOLDEN, "mixed_play_cases.npz"))
    for c in range(int(z["n_cases"])):
        yield int(z[f"c{c}_L"]), {k[len(f"c{c}_"):]: z[k] for k in z.files if k.startswith(f"c{c}_")}


def test_oracle_reproduces_the_reference_run():
    for L, g in golden():
        acted = (g["joint_actions"][..., 0] == 7).astype(np.int32)           # marker: the frozen convention's actor answers 7
        for step in range(2 * L):
            for side in range(2):
                assert np.array_equal(policy_choice(L, step, g["draws"][step, side]), acted[step, side]), (L, step, side)
        src = buffer_source_steps(L)
        assert np.array_equal(src, g["obs"][:L, :, 0, 0, 0, 0].astype(np.int64))    # obs[..., 0] = wall step that produced it
        assert src.min() >= 1 and src.max() == 2 * L - 1


def test_schedule_matches_reference_choice_and_placement():
    for L, g in golden():
        s = MixedPlaySchedule(L, L - 1)
        acted = (g["joint_actions"][..., 0] == 7).astype(np.int32)
        for step in range(2 * L):
            ch = s.choose(step, torch.from_numpy(g["draws"][step]))             # [2, L-1]
            assert np.array_equal(ch.numpy(), acted[step])
        assert np.array_equal(s.choose_all(torch.from_numpy(g["draws"])).numpy(), acted)
        assert np.array_equal(s.source_step().numpy(), g["obs"][:L, :, 0, 0, 0, 0].astype(np.int64))
        # the whole reference buffer is one gather out of time-aligned per-step data: rebuild every stored quantity
        # from what each (wall step, env, agent) produced in the reference run
        idx = s.gather_index(2).view(L, 2, L - 1)
        step_of = idx // (2 * (L - 1))
        agent_of, env_of = (idx // (L - 1)) % 2, idx % (L - 1)
        assert np.array_equal(agent_of.numpy(), np.broadcast_to(np.arange(2)[None, :, None], (L, 2, L - 1)))
        for t in range(L):
            for a in range(2):
                for e in range(L - 1):
                    w = int(step_of[t, a, e])
                    assert int(env_of[t, a, e]) == e
                    assert g["obs"][t, e, a, 0, 0, 0] == w and g["obs"][t, e, a, 0, 0, 1] == e and g["obs"][t, e, a, 0, 0, 2] == a
                    assert g["actions"][t, e, a, 0] == 0                         # only forced (trained-policy) steps are kept
                    assert g["rewards"][t, e, a, 0] == 100.0 * w + e
                    assert g["value_preds"][t, e, a, 0] == 1000.0 + w            # ... valued by the trained policy's own critic


def test_schedule_many_kitchens_repeats_the_staircase():
    L, reps = 7, 5
    s = MixedPlaySchedule(L, reps * (L - 1))
    want = buffer_source_steps(L)
    assert np.array_equal(s.source_step().numpy(), np.repeat(want, reps, axis=1))       # kitchen k plays environment k // reps
    u = torch.rand(2, reps * (L - 1))
    for step in range(2 * L):
        ch = s.choose(step, u).numpy()
        for r in range(reps):                                   # every reps-th kitchen: one full copy of the reference's envs
            for side in range(2):
                assert np.array_equal(ch[side, r::reps], policy_choice(L, step, u[side, r::reps].numpy()))


@pytest.mark.gpu
def test_mixed_play_rollout_on_the_engine(layout_params):
    """Per-row policy mixing in ONE policy launch per step, the env writing the trajectory in place, and the mixed-play
    buffer as one gather: checked against per-network PyTorch evaluation, the oracle env and the oracle placement."""
    from diverse_conventions_amd import OvercookedEngine, PackedNetworks
    from diverse_conventions_amd.mixed_play import MixedPlayRollout
    from torch_policies import StackedActors, StackedCritics
    from oracle import oracle
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    pick = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and ("weight" in k or "bias" in k)}
    actors, critics = [pick("a0."), pick("a1.")], [pick("c0."), pick("c1.")]
    L, reps = 12, 9
    n = reps * (L - 1)
    params = dict(layout_params["simple"], horizon=L)
    eng = OvercookedEngine(params, n, 0)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=actors, critics=critics)
    ro = MixedPlayRollout(eng, nets, L, main=(0, 2), partner=(1, 3), deterministic=True)
    g = torch.Generator(device="cuda").manual_seed(5)
    uniforms = torch.rand((2 * L, 2, n), device="cuda", generator=g)
    ro.collect(uniforms)
    sa, sc = StackedActors(actors, (5, 4, 20), "cuda"), StackedCritics(critics, (5, 4, 20), "cuda")
    ora = oracle.OracleVecEnv(params, n)
    assert np.array_equal(ro.traj.obs[0].cpu().numpy(), ora.encode())
    acts = ro.traj.actions.cpu().numpy()
    for i in range(2 * L):
        ch = ro.choice[i]
        for side in range(2):
            want = policy_choice(L, i, uniforms[i, side, ::reps].cpu().numpy())
            assert np.array_equal(ch[side, ::reps].cpu().numpy(), want)
            obs = ro.traj.obs[i, side].unsqueeze(0)
            a = torch.stack([sa.act(obs, torch.tensor([k]), deterministic=True)[0].view(-1) for k in range(2)])
            v = torch.stack([sc.values(obs, torch.tensor([k])).view(-1) for k in range(2)])
            sel = ch[side].long().unsqueeze(0)
            assert np.array_equal(a.gather(0, sel)[0].cpu().numpy(), acts[i, side])
            assert (v.gather(0, sel)[0] - ro.values[i, side]).abs().max() < 2e-6
        oobs, orew, odone = ora.step(acts[i])
        assert np.array_equal(ro.traj.obs[i + 1].cpu().numpy(), oobs) and np.array_equal(ro.traj.done[i].cpu().numpy(), odone)
        assert int(odone.sum()) == (n if i in (L - 1, 2 * L - 1) else 0)
    buf = ro.buffer()
    src = np.repeat(buffer_source_steps(L), reps, axis=1)                # [L, N]: kitchen k plays environment k // reps
    assert np.array_equal(buf["source_step"].cpu().numpy(), src)
    tobs, tact = ro.traj.obs.cpu().numpy(), acts
    for t in range(L):
        for k in range(0, n, 7):
            w = src[t, k]
            assert np.array_equal(buf["obs"][t, :, k].cpu().numpy(), tobs[w, :, k])
            assert np.array_equal(buf["actions"][t, :, k].cpu().numpy(), tact[w, :, k])
            assert np.array_equal(buf["value_preds"][t, :, k].cpu().numpy(), ro.values[w, :, k].cpu().numpy())
            assert float(buf["masks"][t, 0, k]) == (0.0 if w in (L - 1, 2 * L - 1) else 1.0)
    assert int(buf["choice"].abs().sum()) == 0                           # every kept step was played by the trained policy
    mean_ret, count = ro.traj.episode_returns()                          # two episodes of L steps ended in the 2L-step buffer
    rew = ro.traj.reward[:, 0].cpu().numpy().astype(np.float64)
    assert np.array_equal(count.cpu().numpy(), np.full(n, 2))
    assert np.allclose(mean_ret.cpu().numpy(), rew.sum(0) / 2)
    eng.close()


@pytest.mark.gpu
def test_two_collections_draw_fresh_action_noise(layout_params):
    """The uniform behind a sampled action is hash(seed, draw counter + t, row, 0): a second collect() with the same
    weights, the same coin flips and the same seed must continue the counter, not replay the first collection's noise."""
    from diverse_conventions_amd import OvercookedEngine, PackedNetworks
    from diverse_conventions_amd.fused_policy import draw_uniform
    from diverse_conventions_amd.mixed_play import MixedPlayRollout
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    pick = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and ("weight" in k or "bias" in k)}
    L, n = 6, 40
    eng = OvercookedEngine(dict(layout_params["simple"], horizon=L), n, 0)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[pick("a0."), pick("a1.")], critics=[pick("c0."), pick("c1.")])
    ro = MixedPlayRollout(eng, nets, L, main=(0, 2), partner=(1, 3), seed=77)
    uniforms = torch.rand((2 * L, 2, n), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    a1 = ro.collect(uniforms).traj.actions.clone()
    lp1 = ro.logp.clone()
    assert ro.draws_done == 2 * L
    a2 = ro.collect(uniforms).traj.actions.clone()
    assert ro.draws_done == 4 * L
    # step 0 sees the same start observation and the same policy choice in both collections: only the noise can differ
    assert not torch.equal(a1[0], a2[0]), "the second collection replayed the first one's sampling noise"
    assert torch.equal(lp1[0].isfinite(), torch.ones_like(lp1[0], dtype=torch.bool))
    # and the draw of step t of the second collection is the one the exported hash gives for counter 2L + t
    u_first = np.array([draw_uniform(77, 0, r, a) for r in range(4) for a in range(6)])
    u_second = np.array([draw_uniform(77, 2 * L, r, a) for r in range(4) for a in range(6)])
    assert not np.array_equal(u_first, u_second)
    eng.close()
