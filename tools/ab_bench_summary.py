#!/usr/bin/env python3
"""One line of the env-kernel figures of a bench.py JSON line (for tools/ab_builds.sh A/B runs of two libraries):
    tools/ab_bench_summary.py <log of a bench.py run>      -- parse that log's last JSON line
    tools/ab_bench_summary.py                              -- run `bench.py --no-policy --no-cpu-baseline` now and summarise it"""
import json, os, subprocess, sys
if len(sys.argv) > 1:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
else:
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--no-policy", "--no-cpu-baseline"],
                       capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
out = {"headline_us": round(d["roofline"]["us_per_launch"], 3), "hbm_proper_us": round(d["roofline_hbm_proper"]["us_per_launch"], 3),
       "config4_split_us": round(d["config4_split"]["us_per_launch"], 3), "config3_one_launch_us": round(d["config3_one_launch"]["us_per_joint_step"], 3),
       "config2_us": round(d["config2"]["us_per_launch"], 3), "fused_us": round(d["fused_rollout"]["us_per_step"], 3)}
if "policy_in_loop" in d and "error" not in d["policy_in_loop"]:
    out["policy_in_loop_us"] = [round(x, 2) for x in d["policy_in_loop"]["us_per_step_by_mode"].values()]
    out["policy_kernel_us"] = round(d["policy_in_loop"]["policy_kernel"]["us_per_launch"], 2)
if "mixed_play" in d and "error" not in d["mixed_play"]:
    out["mixed_play_us"] = round(d["mixed_play"]["us_per_step"], 2)
if "config4_share_policy_in_loop" in d and "error" not in d["config4_share_policy_in_loop"]:
    out["config4_share_us"] = round(d["config4_share_policy_in_loop"]["us_per_step"], 2)
print(json.dumps(out))
