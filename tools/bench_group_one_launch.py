#!/usr/bin/env python3
"""The multi-grid policy launch + ONE env launch per joint step (the default of GroupPairedRollout) at config 3's and config 5's
per-GPU shapes: the three `one_launch` lines of `bench_configs.py --group-policies` without the slower alternatives."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import config3_group_policies

for kw in (dict(critics=False), dict(critics=True), dict(k=31, critics=True)):
    print(json.dumps(config3_group_policies(partition="one_launch", **kw)), flush=True)
