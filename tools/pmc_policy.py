#!/usr/bin/env python3
"""Workload for a `rocprofv3 --pmc` pass over the fused policy kernel alone: 8 conventions, 15 segments of 4096
kitchens, actors + critics for both sides (245 760 network evaluations per launch), 6 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_fused_policy import run
layout = sys.argv[1] if len(sys.argv) > 1 else "cramped_room"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
print(run(layout, threads, 7, iters=3, with_torch=False))
