"""Run by tests/test_bench_multi.py through the clean launcher under torch.distributed.run --nproc-per-node 1: the RCCL
("nccl") code paths of the package with one rank on the one GPU of the test box -- process-group init bound to the
device, the packed trajectory all-gather as a real RCCL call on device buffers, the score all-reduce."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from diverse_conventions_amd import OvercookedEngine, TrajectoryExchange, get_base_layout_params, pairing_segments
from diverse_conventions_amd.sharding import segment_scores

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
params = get_base_layout_params("coordination_ring", 200)
n, T = 384, 12
eng = OvercookedEngine(params, n, 0)
r4 = eng.record_bytes // 4
rec = torch.zeros((T, n, r4), dtype=torch.int32, device="cuda")
rew = torch.zeros((T, n), dtype=torch.int32, device="cuda")
done = torch.zeros((T, n), dtype=torch.int32, device="cuda")
acts = torch.randint(0, 6, (T, 2, n), device="cuda", dtype=torch.int32)
obs = torch.zeros((T,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
for t in range(T):
    eng.export_states(out=rec[t])
    obs[t].copy_(eng.obs)
    _, r, d = eng.step(acts[t])
    rew[t].copy_(r[0]); done[t].copy_(d)
ex = TrajectoryExchange(always_collective=True)
got = ex.gather_trajectory(rec, acts, rew, done)
ok = ex.collectives == 1 and got["records"].shape == (1, T, n, r4)
ok &= bool(torch.equal(got["records"][0], rec) and torch.equal(got["actions"][0], acts) and torch.equal(got["reward"][0], rew))
ok &= bool(torch.equal(ex.featurize_all(eng, got["records"]), obs))
segs = pairing_segments(n // 3, 1)
scores = segment_scores(torch.arange(n, device="cuda", dtype=torch.float32), segs)
ok &= bool(torch.allclose(scores.cpu(), torch.tensor([(n // 3 - 1) / 2 + i * (n // 3) for i in range(3)])))
print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "world": dist.get_world_size(), "collectives": ex.collectives}))
eng.close()
dist.destroy_process_group()
