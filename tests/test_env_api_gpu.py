"""Drop-in surface on the GPU: OvercookedMadrona / generate_env driven exactly the way the reference
runners drive it (train/MAPPO/main_player.py:211-277), checked against the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu


class ScriptedPartner:
    """VectorAgent that plays a pre-drawn action table and records what it is told."""
    def __init__(self, table):
        self.table, self.t, self.updates = table, 0, []

    def get_action(self, obs, record=True):
        a = self.table[self.t]
        self.t += 1
        return a

    def update(self, rewards, dones):
        self.updates.append((rewards.clone(), dones.clone()))


@pytest.mark.parametrize("ego", [0, 1])
def test_overcooked_madrona_drop_in(ego, layout_params):
    from diverse_conventions_amd import OvercookedMadrona, VectorObservation, generate_env
    n, T = 50, 230
    env = generate_env("overcooked", n, "simple") if ego == 0 else OvercookedMadrona("cramped_room", n, 0, debug_compile=False, ego_agent_idx=1)
    assert env.num_envs == n and env.n_players == 2 and env.ego_ind == ego
    assert tuple(env.observation_space.shape) == (5, 4, 20) and env.action_space.n == 6
    assert env.action_space.__class__.__name__ == "Discrete"
    assert env.share_observation_space is env.observation_space
    rng = np.random.default_rng(ego)
    acts = rng.integers(0, 6, size=(T, 2, n, 1))
    dacts = torch.from_numpy(acts).cuda()                  # int64, like sampled torch actions
    partner = ScriptedPartner(dacts[:, 1 - ego])
    env.add_partner_agent(partner, player_num=1 - ego)
    ora = oracle.OracleVecEnv(layout_params["simple"], n)

    vobs = env.reset()
    assert isinstance(vobs, VectorObservation)
    assert vobs.obs.shape == (n, 5, 4, 20) and vobs.obs.dtype == torch.int8
    assert vobs.state is vobs.obs and vobs.action_mask.shape == (n, 6) and vobs.action_mask.dtype == torch.bool
    assert bool(vobs.active.all()) and bool(vobs.action_mask.all())
    assert np.array_equal(vobs.obs.cpu().numpy(), ora.encode()[ego])
    for t in range(T):
        vobs, rew, done, info = env.step(dacts[t, ego])
        oobs, orew, odone = ora.step(acts[t, :, :, 0])
        assert np.array_equal(vobs.obs.clone().cpu().numpy(), oobs[ego]), t
        assert rew.shape == (n,) and rew.dtype == torch.int32 and np.array_equal(rew.cpu().numpy(), orew[ego])
        assert done.shape == (n,) and np.array_equal(done.to(torch.bool).cpu().numpy(), odone.astype(bool))
        assert isinstance(info, list) and len(info) == n
    assert len(partner.updates) == T
    assert np.array_equal(partner.updates[-1][0].cpu().numpy(), orew[1 - ego])
    # n_reset re-reads, it does not restart running episodes (reference envs/overcooked2_env.py:127-128)
    again = env.reset()
    assert np.array_equal(again.obs.cpu().numpy(), oobs[ego])
    env.close()


def test_env_tensors_on_cpu_when_asked(layout_params):
    from diverse_conventions_amd import generate_env
    env = generate_env("overcooked", 8, "unident_s", use_env_cpu=True)
    from diverse_conventions_amd import RandomVectorAgent
    env.add_partner_agent(RandomVectorAgent(lambda: torch.randint(0, 6, (8, 1))))
    vobs = env.reset()
    assert vobs.obs.device.type == "cpu" and vobs.obs.shape == (8, 9, 5, 20)
    vobs, rew, done, _ = env.step(torch.randint(0, 6, (8, 1)))
    assert rew.device.type == "cpu" and done.device.type == "cpu"
    env.close()


def test_constructor_errors():
    from diverse_conventions_amd import OvercookedMadrona
    from diverse_conventions_amd.native import OvercookedNativeError
    with pytest.raises(RuntimeError):
        OvercookedMadrona("simple", 4, 0, use_cpu=True)
    with pytest.raises(FileNotFoundError):
        OvercookedMadrona("no_such_kitchen", 4, 0)
    with pytest.raises(OvercookedNativeError):
        OvercookedMadrona("multiplayer_schelling", 0, 0)      # no kitchens: reported, not crashed


def test_fused_partner_agent_in_the_reference_loop(layout_params):
    """The reference-style loop (env.add_partner_agent / env.step(ego actions)) with the partner's conventions on the
    fused policy kernel: three segments played by different actors, checked against per-segment PyTorch evaluation of
    the same reference-format weights and the oracle env."""
    import os
    from conftest import GOLDEN
    from diverse_conventions_amd import FusedVectorAgent, PackedNetworks, generate_env
    from torch_policies import StackedActors, StackedCritics
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    pick = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and ("weight" in k or "bias" in k)}
    actors, critics = [pick(f"a{i}.") for i in range(3)], [pick("c0.")]
    n, T = 150, 60
    env = generate_env("overcooked", n, "simple")
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=actors, critics=critics)
    segments = [(0, 50, 2, 3), (50, 50, 0, 3), (100, 50, 1, 3)]
    partner = FusedVectorAgent(nets, n, segments, deterministic=True)
    env.add_partner_agent(partner)
    sa, sc = StackedActors(actors, (5, 4, 20), "cuda"), StackedCritics(critics, (5, 4, 20), "cuda")
    ora = oracle.OracleVecEnv(layout_params["simple"], n)
    rng = np.random.default_rng(3)
    ego = torch.from_numpy(rng.integers(0, 6, size=(T, n, 1))).cuda()
    env.reset()
    for t in range(T):
        pobs = env.static_observations[1].clone()                 # what the partner is about to see
        vobs, rew, done, _ = env.step(ego[t])
        want = torch.cat([sa.act(pobs[f:f + c].unsqueeze(0), torch.tensor([a]), deterministic=True)[0].view(-1) for f, c, a, _ in segments])
        assert np.array_equal(partner.actions.cpu().numpy(), want.cpu().numpy()), t
        assert (sc.values(pobs.unsqueeze(0), torch.tensor([0])).view(-1) - partner.values).abs().max() < 2e-6
        joint = np.stack([ego[t, :, 0].cpu().numpy(), partner.actions.cpu().numpy()])
        oobs, orew, odone = ora.step(joint)
        assert np.array_equal(vobs.obs.cpu().numpy(), oobs[0]) and np.array_equal(rew.cpu().numpy(), orew[0])
    env.close()


@pytest.mark.parametrize("ego_ind", [0, 1])
def test_agent_pair_evaluates_both_players_in_one_launch(layout_params, ego_ind):
    """FusedAgentPair in the reference's evaluation loop (train/testing.py:39-59: ego.get_action, then env.step asks the partner): ONE
    policy launch per step over both players' halves of the simulator's observation tensor -- same actions, log-probabilities and
    values as two FusedVectorAgents in a second environment driven side by side (deterministic policies), the env against the oracle;
    and the fallbacks: an observation that is no view of the [2, N, ...] tensor, the partner asked first."""
    import os
    from conftest import GOLDEN
    from diverse_conventions_amd import FusedAgentPair, FusedVectorAgent, OvercookedMadrona, PackedNetworks, VectorObservation
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    pick = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and ("weight" in k or "bias" in k)}
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[pick(f"a{i}.") for i in range(3)], critics=[pick("c0.")])
    n, T = 333, 40
    ego_segs, alt_segs = [(0, 200, 0, 3), (200, 133, 1, 3)], [(0, n, 2, 3)]
    env_a, env_b = (OvercookedMadrona("simple", n, 0, debug_compile=False, ego_agent_idx=ego_ind) for _ in range(2))
    pair = FusedAgentPair(nets, n, ego_segs, alt_segs, ego_ind=ego_ind, deterministic=True)
    env_a.add_partner_agent(pair.partner, player_num=1 - ego_ind)
    ego_b, alt_b = FusedVectorAgent(nets, n, ego_segs, deterministic=True), FusedVectorAgent(nets, n, alt_segs, deterministic=True)
    env_b.add_partner_agent(alt_b, player_num=1 - ego_ind)
    ora = oracle.OracleVecEnv(layout_params["simple"], n)
    oa, ob = env_a.reset(), env_b.reset()
    for t in range(T):
        a = pair.ego.get_action(oa)
        b = ego_b.get_action(ob)
        assert torch.equal(a, b), t
        oa, ra, da, _ = env_a.step(a)
        ob, rb, db, _ = env_b.step(b)
        assert torch.equal(pair.partner.actions, alt_b.actions) and torch.equal(pair.partner.action_logp, alt_b.action_logp), t
        assert torch.equal(pair.ego.values, ego_b.values) and torch.equal(pair.partner.values, alt_b.values), t
        assert torch.equal(oa.obs, ob.obs) and torch.equal(ra, rb) and torch.equal(da, db), t
        joint = [a[:, 0].cpu().numpy(), pair.partner.actions.cpu().numpy()]
        oobs, orew, odone = ora.step(np.stack(joint if ego_ind == 0 else joint[::-1]))
        assert np.array_equal(oa.obs.cpu().numpy(), oobs[ego_ind]) and np.array_equal(ra.cpu().numpy(), orew[ego_ind]), t
    assert pair.joint_launches == T and pair.single_launches == 0
    # fallbacks: a copy of the ego's view (no view of the whole tensor) -> the ego's rows alone, then the partner's alone; the partner first
    copy = VectorObservation(oa.active, oa.obs.clone(), action_mask=oa.action_mask)
    a1 = pair.ego.get_action(copy).clone()
    assert pair.single_launches == 1 and torch.equal(a1, ego_b.get_action(ob))
    p1 = pair.partner.get_action(env_a._obs[1 - ego_ind]).clone()
    assert pair.single_launches == 2 and torch.equal(p1, alt_b.get_action(env_b._obs[1 - ego_ind]))
    env_a.close(); env_b.close()


def test_the_simulator_stub_shown_in_integration_md_runs(layout_params):
    """INTEGRATION.md section 2 is the ctypes stand-in for the reference's nanobind module
    (build.madrona_simplecooked_example_python.SimplecookedSimulator, src/overcooked2_env/bindings.cpp:22-98): execute that
    block as written and drive it the way OvercookedMadrona does -- write the actions in place, step(), read obs / reward /
    done from the same persistent tensors (envs/overcooked2_env.py:116-125) -- against the oracle."""
    import os
    import re
    from conftest import ROOT
    from diverse_conventions_amd import native
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n# envs/overcooked_hip_binding\.py\n(.*?)```", text, re.S).group(1)
    block = block.replace('"diverse-conventions_amd/csrc/liboc_hip.so"', repr(native.LIB_PATH))
    ns = {}
    exec(compile(block, "INTEGRATION.md:2", "exec"), ns)
    params = layout_params["random1"]
    n, T = 300, 205
    kw = {k: params[k] for k in ("terrain", "height", "width", "num_players", "start_player_x", "start_player_y", "placement_in_pot_rew",
                                 "dish_pickup_rew", "soup_pickup_rew", "recipe_values", "recipe_times", "horizon")}
    sim = ns["SimplecookedSimulator"](exec_mode=None, gpu_id=0, num_worlds=n, debug_compile=False, **kw)     # bindings.cpp:28-84
    ora = oracle.OracleVecEnv(params, n)
    assert np.array_equal(sim.obs.cpu().numpy(), ora.encode())
    rng = np.random.default_rng(4)
    acts = rng.integers(0, 6, size=(T, 2, n, 1)).astype(np.int32)
    for t in range(T):
        sim.actions.copy_(torch.from_numpy(acts[t]).cuda())          # static_actions.copy_(...), overcooked2_env.py:119
        sim.step()                                                   # self.sim.step(), :121
        oobs, orew, odone = ora.step(acts[t][:, :, 0])
        assert np.array_equal(sim.obs.cpu().numpy(), oobs), t
        assert np.array_equal(sim.reward.cpu().numpy(), orew) and np.array_equal(sim.done.cpu().numpy(), odone)
    assert int(ora.timestep[0]) == T - 200                           # the horizon reset happened inside step 200
