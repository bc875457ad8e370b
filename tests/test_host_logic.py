"""Host-side logic that needs no GPU: state (de)serialisation and the ego-view vector-env plumbing
(own implementation of the reference's pantheonrl_extension/vectorenv.py surface)."""
import numpy as np
import pytest
import torch

from diverse_conventions_amd import (PlayerException, RandomVectorAgent, VectorAgent, VectorMultiAgentEnv,
                                     VectorObservation)
from diverse_conventions_amd.engine import pack_states, unpack_states
from oracle import oracle


def test_state_pack_roundtrip(layout_params):
    params = layout_params["unident_s"]
    env = oracle.OracleVecEnv(params, 300, nthreads=2)
    rng = np.random.default_rng(0)
    for _ in range(150):
        env.step(rng.integers(0, 6, size=(2, 300)))
    tiles, players, ts = env.get_state()
    rec_bytes = 16 * ((45 + 4 + 3) // 4)
    rec = pack_states(tiles, players, ts, rec_bytes)
    assert rec.shape == (300, rec_bytes // 4) and rec.dtype == np.uint32
    t2, p2, ts2 = unpack_states(rec, 45)
    assert np.array_equal(t2, tiles) and np.array_equal(p2, players) and np.array_equal(ts2, ts)
    # documented packing: name | onions<<8 | tomatoes<<16 | tick<<24 ; posor bytes
    k = int(np.argmax(tiles[:, :, 0].max(axis=1) > 0))
    t = int(np.argmax(tiles[k, :, 0] > 0))
    name, on, to, tick = (int(x) for x in tiles[k, t])
    assert rec[k, t] == (name | on << 8 | to << 16 | (tick & 0xFF) << 24)
    assert rec[k, 45] == (int(players[k, 0, 0]) | int(players[k, 0, 1]) << 8 | int(players[k, 1, 0]) << 16 | int(players[k, 1, 1]) << 24)


class CountingEnv(VectorMultiAgentEnv):
    """obs = step counter; reward = sum of the two actions; done every 3rd step."""
    def __init__(self, n, ego=0):
        super().__init__(n, torch.device("cpu"), ego_ind=ego, n_players=2)
        self.t = 0

    def _obs_list(self):
        o = torch.full((self.num_envs, 2), float(self.t))
        return [VectorObservation(torch.ones(self.num_envs, dtype=torch.bool), o + i) for i in range(2)]

    def n_step(self, actions):
        self.t += 1
        rew = actions.sum(dim=0).squeeze(-1).repeat(2, 1)
        done = torch.full((self.num_envs,), self.t % 3 == 0)
        return self._obs_list(), rew, done, [{}] * self.num_envs

    def n_reset(self):
        return self._obs_list()


class Recorder(VectorAgent):
    def __init__(self, value, n):
        self.value, self.n, self.seen, self.updates = value, n, [], []

    def get_action(self, obs, record=True):
        self.seen.append(obs.obs.clone())
        return torch.full((self.n, 1), self.value)

    def update(self, rewards, dones):
        self.updates.append((rewards.clone(), dones.clone()))


@pytest.mark.parametrize("ego", [0, 1])
def test_ego_view_step_and_partner_plumbing(ego):
    env = CountingEnv(5, ego)
    a, b = Recorder(2, 5), Recorder(4, 5)
    env.add_partner_agent(a, player_num=1 - ego)
    env.add_partner_agent(b, player_num=1 - ego)
    first = env.reset()                      # round robin: second partner becomes current
    assert env.partnerids == [1] and torch.equal(first.obs, torch.full((5, 2), float(ego)))
    assert first.state is first.obs
    obs, rew, done, info = env.step(torch.full((5, 1), 1))
    assert torch.equal(rew, torch.full((5,), 5)) and not done.any() and len(info) == 5
    assert len(b.seen) == 1 and len(a.seen) == 0 and torch.equal(b.seen[0], torch.full((5, 2), float(1 - ego)))
    assert len(b.updates) == 1 and torch.equal(b.updates[0][0], torch.full((5,), 5))
    env.set_partnerid(0, player_num=1 - ego)
    env.step(torch.full((5, 1), 1)); obs, rew, done, _ = env.step(torch.full((5, 1), 0))
    assert done.all() and torch.equal(rew, torch.full((5,), 2)) and len(a.updates) == 2
    assert torch.equal(obs.obs, torch.full((5, 2), 3.0 + ego))


def test_player_exceptions_and_policies():
    env = CountingEnv(2)
    with pytest.raises(PlayerException):
        env.add_partner_agent(RandomVectorAgent(lambda: None), player_num=0)      # ego slot
    with pytest.raises(PlayerException):
        env.set_resample_policy("bogus")
    with pytest.raises(PlayerException):
        VectorMultiAgentEnv.__init__(env, 2, torch.device("cpu"), n_players=3, resample_policy="robin")
    with pytest.raises(PlayerException):
        VectorMultiAgentEnv.__init__(env, 2, torch.device("cpu"), partners=[[], []])
    env2 = CountingEnv(2)
    env2.set_resample_policy("random")
    for _ in range(3):
        env2.add_partner_agent(RandomVectorAgent(lambda: torch.zeros(2, 1, dtype=torch.long)))
    np.random.seed(0)
    seen = set()
    for _ in range(30):
        env2.reset()
        seen.add(env2.partnerids[0])
    assert seen == {0, 1, 2}
    assert env2.getDummyEnv(1) is env2


def test_vector_observation_defaults():
    o = torch.zeros(3, 4)
    v = VectorObservation(torch.ones(3, dtype=torch.bool), o)
    assert v.state is o and v.action_mask is None
    s = torch.ones(3, 2)
    v = VectorObservation(torch.ones(3, dtype=torch.bool), o, s, torch.ones(3, 6, dtype=torch.bool))
    assert v.state is s and v.action_mask.shape == (3, 6)


def test_bench_refuses_two_ranks_on_one_device():
    """bench.py's exchange leg ends the run when two ranks report the same GPU (the multi-GPU contract is one rank per GPU),
    unless the OC_BENCH_DEVICE test hook pinned them there."""
    import bench
    a = {"rank": 0, "device_index": 0, "uuid": "GPU-aa"}
    b = {"rank": 1, "device_index": 1, "uuid": "GPU-bb"}
    assert bench.check_distinct_devices([a, b]) == 2
    assert bench.check_distinct_devices([dict(a, uuid=None), dict(b, uuid=None)]) == 2
    assert bench.check_distinct_devices([a, dict(b, uuid="GPU-aa")]) == 2          # one uuid reported for two device indices: still two devices
    with pytest.raises(SystemExit, match="one rank per GPU"):
        bench.check_distinct_devices([a, dict(b, device_index=0)])
    assert bench.check_distinct_devices([a, dict(b, device_index=0)], allow_shared=True) == 1
    assert bench.algorithmic_bytes(20) == 1012 and bench.algorithmic_bytes(45) == 2212


def test_bench_self_launch_builds_the_drivers_command(monkeypatch):
    """`python bench.py --gpus N` without a rendezvous in the environment starts torch.distributed.run as a CHILD process with the
    driver's own arguments and relays its line and exit code; the parent imports neither torch.cuda nor the engine."""
    import importlib
    import io
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None, cwd=None):
            seen.update(cmd=cmd, env=env, cwd=cwd)
            self.stdout = io.StringIO('noise\n{"metric": "m", "n_gpus": 4}\n')

        def wait(self):
            return 0

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    out = io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]
    assert seen["env"]["OC_BENCH_LAUNCHER"] == "self" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out.getvalue().count('{"metric"') == 1
    # a child that fails: its exit code is the launcher's
    FakeProc.wait = lambda self: 3
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3
