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
from diverse_conventions_amd import TrajectoryBuffer
traj = TrajectoryBuffer(eng, T, store_states=True)                 # records / actions / reward / done: views of ONE wire buffer
acts = torch.randint(0, 6, (T, 2, n), device="cuda", dtype=torch.int32)
for t in range(T):
    traj.step(acts[t])                                             # one launch per step, the records written by the step kernel
ex = TrajectoryExchange(always_collective=True)
got = ex.gather_buffer(traj)                                       # the wire buffer itself through all_gather_into_tensor (RCCL)
ok = ex.collectives == 1 and got["records"].shape == (1, T + 1, n, eng.record_bytes // 4)
ok &= bool(torch.equal(got["records"][0], traj.records) and torch.equal(got["actions"][0], acts) and torch.equal(got["reward"][0], traj.reward))
ok &= bool(torch.equal(got["done"][0], traj.done) and torch.equal(ex.featurize_all(eng, got["records"]), traj.obs))
# loose fields (one concatenation pass) give the same bytes
loose = ex.gather_trajectory(traj.records[:T].contiguous(), acts, traj.reward[:, 0].contiguous(), traj.done)
ok &= ex.collectives == 2 and bool(torch.equal(loose["records"][0], traj.records[:T]) and torch.equal(loose["done"][0], traj.done))
segs = pairing_segments(n // 3, 1)
scores = segment_scores(torch.arange(n, device="cuda", dtype=torch.float32), segs)
ok &= bool(torch.allclose(scores.cpu(), torch.tensor([(n // 3 - 1) / 2 + i * (n // 3) for i in range(3)])))
print(json.dumps({"ok": bool(ok), "backend": dist.get_backend(), "world": dist.get_world_size(), "collectives": ex.collectives}))
eng.close()
dist.destroy_process_group()
