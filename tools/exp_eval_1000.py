#!/usr/bin/env python3
"""bench.py`s drop_in_step / eval_1000 legs alone (A/B runs): prints the eval_1000 object."""
import sys, json, torch
sys.path.insert(0,'/root/repo')
import bench
class R:
    def __enter__(self): return self
    def __exit__(self,*a): return False
gen = torch.Generator(device="cuda").manual_seed(0)
drop, ev = bench.drop_in_and_eval_legs(0, R(), gen)
print(json.dumps(ev)[:1500])
