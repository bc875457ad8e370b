#!/usr/bin/env python3
"""What the bracket barrier of bench.py's timed region costs with RCCL (one rank is all a 1-GPU box can host; more ranks add the
xGMI hops): dist.barrier() against an all-reduce of one pre-allocated element, each followed by torch.cuda.synchronize().
Run: OC_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/exp_barrier_cost.py"""
import os, time
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
one = torch.zeros(1, device="cuda")
def t(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
def b1(): dist.barrier(); torch.cuda.synchronize()
def b2(): dist.all_reduce(one); torch.cuda.synchronize()
def b3(): dist.barrier(device_ids=[0]); torch.cuda.synchronize()
def b4(): torch.cuda.synchronize()
for name, fn in (("dist.barrier() + sync", b1), ("all_reduce(1 element) + sync", b2), ("dist.barrier(device_ids) + sync", b3), ("sync only", b4)):
    print(f"{name:34s} {t(fn):7.1f} us")
dist.destroy_process_group()
