"""Multi-GPU plumbing on CPU: shard arithmetic, the reference's pairing-segment map, and the compact
trajectory all-gather over a world_size-2 gloo group (RCCL on the GPU box uses the same code path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diverse_conventions_amd.sharding import (KitchenShard, TrajectoryExchange, best_partner, local_segments, pairing_segments,
                                              segment_scores)
from diverse_conventions_amd.engine import pack_states, unpack_states


def test_shards_cover_everything_once():
    for total, world in [(65536, 8), (10, 3), (7, 8), (4099, 4)]:
        shards = [KitchenShard(total, world, r) for r in range(world)]
        assert sum(s.count for s in shards) == total
        assert [s.first for s in shards] == list(np.cumsum([0] + [s.count for s in shards[:-1]]))
        assert shards[0].counts() == [s.count for s in shards]
        assert max(s.count for s in shards) - min(s.count for s in shards) <= 1


def test_aligned_threads():
    from diverse_conventions_amd import aligned_threads
    assert aligned_threads(8192, 15) == 256 and aligned_threads(8192, 31) == 128 and aligned_threads(65536, 7) == 4352
    assert aligned_threads(100, 7) == 6 and aligned_threads(31 * 32, 15) == 32 and aligned_threads(8192, 15, multiple=64) == 256


def test_pairing_segments_match_reference_slicing():
    # reference: envs = threads*(2k+1); sp = [:threads]; xp0_i = [threads*(i+1) : threads*(i+2)];
    # xp1_i = [threads*(i+1+k) : threads*(i+2+k)]   (train/XD/xd_player.py:102-140)
    threads, k = 50, 7
    segs = pairing_segments(threads, k)
    assert len(segs) == 2 * k + 1 and segs[-1].stop == threads * (2 * k + 1)
    assert (segs[0].name, segs[0].start, segs[0].stop, segs[0].ego_policy, segs[0].partner_policy) == ("SP", 0, 50, 0, 0)
    for i in range(k):
        a, b = segs[1 + i], segs[1 + k + i]
        assert (a.start, a.stop) == (threads * (i + 1), threads * (i + 2)) and (a.ego_policy, a.partner_policy) == (0, i + 1)
        assert (b.start, b.stop) == (threads * (i + 1 + k), threads * (i + 2 + k)) and (b.ego_policy, b.partner_policy) == (i + 1, 0)
    # ego / partner policy lists of the reference: [new]*(k+1) + convs ; [new] + convs + [new]*k
    assert [s.ego_policy for s in segs] == [0] * (k + 1) + list(range(1, k + 1))
    assert [s.partner_policy for s in segs] == [0] + list(range(1, k + 1)) + [0] * k
    per_gpu = local_segments(8192 // 15 * 8, 7, 8, 3)
    assert len(per_gpu) == 15 and per_gpu[0].stop == 8192 // 15


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from conftest import GOLDEN
        from oracle import oracle
        with open(os.path.join(GOLDEN, "layout_params.json")) as f:
            params = json.load(f)["random1"]
        total, T = 24, 30
        shard = KitchenShard(total, world, rank)
        rng = np.random.default_rng(5)
        acts_all = rng.integers(0, 6, size=(T, 2, total)).astype(np.int32)
        env = oracle.OracleVecEnv(params, shard.count)                  # this rank's kitchens only
        sl = slice(shard.first, shard.first + shard.count)
        rec_bytes = 16 * ((25 + 4 + 3) // 4)
        recs, rews, dones = [], [], []
        for t in range(T):
            _, r, d = env.step(acts_all[t][:, sl])
            recs.append(pack_states(*env.get_state(), rec_bytes).view(np.int32))
            rews.append(r[0].copy()); dones.append(d.copy())
        ex = TrajectoryExchange()
        got = ex.gather_trajectory(torch.from_numpy(np.stack(recs)), torch.from_numpy(acts_all[:, :, sl].copy()),
                                   torch.from_numpy(np.stack(rews)), torch.from_numpy(np.stack(dones)))
        ok_one = ex.collectives == 1                  # the four fields travel in ONE all-gather
        # every rank must now hold the full batch, peers in rank order, equal to a single-process run
        ref = oracle.OracleVecEnv(params, total)
        ok = ok_one and got["records"].shape == (world, T, shard.count, rec_bytes // 4)
        ok &= got["records"].data_ptr() % 16 == 0 and got["records"][1].data_ptr() % 16 == 0    # uint4-readable by the encode kernel
        for t in range(T):
            _, r, d = ref.step(acts_all[t])
            full = pack_states(*ref.get_state(), rec_bytes).view(np.int32)
            glued = torch.cat([got["records"][p, t] for p in range(world)]).numpy()
            ok &= np.array_equal(glued, full)
            ok &= np.array_equal(torch.cat([got["reward"][p, t] for p in range(world)]).numpy(), r[0])
            ok &= np.array_equal(torch.cat([got["done"][p, t] for p in range(world)]).numpy(), d)
            ok &= np.array_equal(torch.cat([got["actions"][p, t] for p in range(world)], dim=1).numpy(), acts_all[t])
        # the receiver can rebuild peers' observations from the compact records (oracle stands in
        # for the encode kernel here; tests/test_multi_gpu_gpu.py does it with oc_encode_states)
        tiles, players, ts = unpack_states(glued.view(np.uint32), 25)
        chk = oracle.OracleVecEnv(params, total)
        chk.set_state(tiles, players, ts)
        ok &= np.array_equal(chk.encode(), ref.encode())
        # score bookkeeping: per-segment mean returns over both ranks == single-process means
        segs = pairing_segments(2, 2)                                   # 5 segments of 2 kitchens per rank
        torch.manual_seed(0)
        full = torch.rand(2, 10, dtype=torch.float64)                   # [rank, kitchen]
        got_scores = segment_scores(full[rank], segs)
        want = torch.stack([full[:, sg.start:sg.stop].mean() for sg in segs])
        ok &= bool(torch.allclose(got_scores, want))
        ok &= best_partner(got_scores, 2) == int((want[1:3] + want[3:5]).argmax())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_compact_trajectory_all_gather_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {0: True, 1: True}


class _OracleEngine:
    """What TrajectoryExchange needs of an engine, served by the CPU oracle: geometry + `encode_trajectory_into` with the engine's
    argument meaning (records [T, n, rec/4] of one peer -> kitchens [offset, offset + n) of every slot of [T, P, N_total, W, H, C])."""

    def __init__(self, params, n):
        from oracle import oracle
        self.params, self.num_kitchens, self.num_players = params, n, 2
        self.width, self.height, self.channels = params["width"], params["height"], 20
        self.size = self.width * self.height
        self.record_bytes = 16 * ((self.size + 4 + 3) // 4)
        self._ora = oracle.OracleVecEnv(params, n)

    def encode_trajectory_into(self, records, obs_global, kitchen_offset):
        T, n = records.shape[0], records.shape[1]
        for t in range(T):
            self._ora.set_state(*unpack_states(records[t].numpy().view(np.uint32), self.size))
            obs_global[t, :, kitchen_offset:kitchen_offset + n] = torch.from_numpy(self._ora.encode())
        return obs_global


def _worker_world8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from types import SimpleNamespace
        from conftest import GOLDEN
        from oracle import oracle
        from diverse_conventions_amd.trajectory import wire_layout
        with open(os.path.join(GOLDEN, "layout_params.json")) as f:
            params = json.load(f)["unident_s"]                      # asymmetric_advantages: config 4's layout
        total, T = world * 6, 9                                      # six kitchens per rank
        shard = KitchenShard(total, world, rank)
        n, sl = shard.count, slice(shard.first, shard.first + shard.count)
        rng = np.random.default_rng(11)
        acts_all = rng.integers(0, 6, size=(T, 2, total)).astype(np.int32)
        eng = _OracleEngine(params, n)
        # this rank's trajectory, laid out exactly as TrajectoryBuffer(store_states=True) lays it out: ONE flat int32 buffer
        fields, words = wire_layout(T, 2, n, eng.record_bytes // 4)
        wire = torch.zeros(words, dtype=torch.int32)
        view = lambda name: wire[fields[name][0]:fields[name][0] + int(np.prod(fields[name][1]))].view(fields[name][1])
        env = oracle.OracleVecEnv(params, n, nthreads=1)
        env.timestep[:] = (np.arange(shard.first, shard.first + n) * 37) % 200       # staggered phases: resets inside the trajectory
        own_obs = [env.encode().copy()]
        view("records")[0] = torch.from_numpy(pack_states(*env.get_state(), eng.record_bytes).view(np.int32))
        for t in range(T):
            o, r, d = env.step(acts_all[t][:, sl])
            own_obs.append(o.copy())
            view("records")[t + 1] = torch.from_numpy(pack_states(*env.get_state(), eng.record_bytes).view(np.int32))
            view("actions")[t] = torch.from_numpy(acts_all[t][:, sl].copy())
            view("reward")[t] = torch.from_numpy(r)
            view("done")[t] = torch.from_numpy(d)
        traj = SimpleNamespace(T=T, P=2, wire=wire, engine=eng)
        ex = TrajectoryExchange()
        recv = torch.empty(world * words, dtype=torch.int32)
        got = ex.gather_buffer(traj, out=recv)
        ok = ex.collectives == 1 and got["records"].shape == (world, T + 1, n, eng.record_bytes // 4) and got["records"].data_ptr() == recv.data_ptr()
        ok &= all(got[k][p].data_ptr() % 16 == 0 for k in got for p in range(world))        # every peer's every field 16-byte aligned
        ok &= all(torch.equal(got[k][rank], view(k)) for k in got)                          # own shard at its own index
        ok &= bool(torch.equal(torch.cat([got["actions"][p] for p in range(world)], dim=2), torch.from_numpy(acts_all)))
        # the receive side: all eight peers' shards re-featurised at their kitchen offsets of the global tensor
        out = ex.featurize_all(eng, got["records"])
        ok &= out.shape == (T + 1, 2, total, eng.width, eng.height, 20)
        ok &= bool(np.array_equal(out[:, :, sl].numpy(), np.stack(own_obs)))
        # ... and every rank holds the same global tensor and the same done / reward columns (checked through a second collective)
        digest = torch.tensor([int(out.to(torch.int64).sum()), int((out.to(torch.int64) * torch.arange(out.numel()).view(out.shape).remainder(8191)).sum()),
                               int(got["done"].sum()), int(got["reward"].sum())])
        all_d = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(all_d, digest)
        ok &= all(torch.equal(d, digest) for d in all_d) and (world < 8 or int(got["done"].sum()) > 0)     # (48 staggered kitchens: some reset inside the trajectory)
        # the same exchange in chunks (ChunkedGather: what overlaps with the rollout on the GPU): chunks of 4, 4 and 1 steps, each chunk's
        # record slots through its own all-gather and re-featurised as it lands, the small fields in one more -- bit-equal to the one-shot
        # gather in the one-shot layout, to its featurize_all, and chunks + 1 collectives
        for chunk_steps in (4, 9, 1):
            ex2 = TrajectoryExchange()
            out2 = torch.zeros_like(out)
            cg = ex2.chunked(traj, chunk_steps, engine=eng, out=out2)
            for c in range(cg.n_chunks):
                cg.submit(c)
            res = cg.finish()
            ok &= ex2.collectives == cg.n_chunks + 1 and cg.n_chunks == -(-T // chunk_steps)
            ok &= bool(torch.equal(cg.records_all(), got["records"])) and bool(torch.equal(out2, out))
            ok &= all(bool(torch.equal(res[k], got[k])) for k in ("actions", "reward", "done"))
            ok &= [v.shape[1] for v in res["records"]] == [s1 - s0 for s0, s1 in cg.slots] and sum(v.shape[1] for v in res["records"]) == T + 1
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_wire_buffer_all_gather_gloo_world8():
    """Eight ranks -- the node the north star names -- on the CPU: every rank's flat wire buffer through ONE all_gather_into_tensor
    (no concatenation), per-field views at eight peers' offsets, and featurize_all dropping eight shards at their kitchen offsets."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_world8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {r: True for r in range(world)}


def test_chunked_exchange_gloo_world2():
    """World 2 over gloo: the chunked exchange (ChunkedGather) against the one-shot gather of the same wire buffers."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_world8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == {r: True for r in range(world)}
