#!/usr/bin/env python3
"""One line of the env-kernel figures of `bench.py --no-policy --no-cpu-baseline` (for tools/ab_builds.sh A/B runs of two libraries)."""
import json, os, subprocess, sys
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--no-policy", "--no-cpu-baseline"],
                   capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print(json.dumps({"headline_us": round(d["roofline"]["us_per_launch"], 3), "hbm_proper_us": round(d["roofline_hbm_proper"]["us_per_launch"], 3),
                  "config4_split_us": round(d["config4_split"]["us_per_launch"], 3), "config3_one_launch_us": round(d["config3_one_launch"]["us_per_joint_step"], 3),
                  "config2_us": round(d["config2"]["us_per_launch"], 3), "fused_us": round(d["fused_rollout"]["us_per_step"], 3)}))
