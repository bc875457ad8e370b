#!/usr/bin/env python3
"""Kitchens per pairing segment as a multiple of 32: no ragged 32-row block at the end of every (segment, network) row range.
Config 4's per-GPU share with 264 kitchens per segment (8184 kitchens) and with 256 (7936), config 5's per-layout share with 130
(8190) and 128 (8064); us per step with the policies (actors + critics) in the loop and agent-env steps/s."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_paired_rollout import run_fused
for layout, threads, k in (("asymmetric_advantages", 264, 15), ("asymmetric_advantages", 256, 15), ("cramped_room", 130, 31), ("cramped_room", 128, 31),
                           ("cramped_room", 4369, 7), ("cramped_room", 4352, 7)):
    r = run_fused(layout, threads, k, critics=True)
    print(json.dumps({"layout": layout, "kitchens_per_segment": threads, "kitchens": r["kitchens"], "segments": r["segments"],
                      "us_per_step": r["enqueued_from_c_us_per_step"], "agent_env_steps_per_s": r["enqueued_from_c_agent_env_steps_per_s"]}), flush=True)
