#!/usr/bin/env python3
"""gpurun_out/<tag>_* (written by tools/profile_round.sh <tag>) -> profiles/: copies the summaries to be judged, refreshes
profiles/kernel_us.json (rocprofv3 average of the headline kernel) and profiles/hbm_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE, the
guide's gfx950 correction), keeps the policy kernel's counter rows only, and prints the lines for profiles/README.md.
    python tools/publish_profiles.py r05z "round 5, final code" """
import csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, label = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for name in ("bench_n1.log", "bench_under_rocprof.log", "bench_under_rocprof_with_secondaries.log", "kernel_stats.csv",
             "kernel_stats_with_secondaries.csv", "pmc_fetch_size.csv", "pmc_write_size.csv"):
    shutil.copy(os.path.join(G, f"{tag}_{name}"), os.path.join(P, f"{tag}_{name}"))
# the policy kernel's rows of the counter pass (the csv also holds every copy / fill kernel of the process)
with open(os.path.join(G, f"{tag}_pmc_fused_policy_all.csv")) as f, open(os.path.join(P, f"{tag}_pmc_fused_policy.csv"), "w", newline="") as g:
    r = csv.reader(f)
    w = csv.writer(g, quoting=csv.QUOTE_NONNUMERIC)
    header = next(r)
    w.writerow(header)
    ki = header.index("Kernel_Name")
    rows = [row for row in r if "oc_policy_kernel" in row[ki] or "oc_policy_duo_kernel" in row[ki]]
    w.writerows(rows)
counters = {}
di, ci, vi = header.index("Dispatch_Id"), header.index("Counter_Name"), header.index("Counter_Value")
for row in rows:
    counters.setdefault(row[ci], []).append(float(row[vi]))
# kernel_us.json
with open(os.path.join(P, f"{tag}_kernel_stats.csv")) as f:
    stats = {row["Name"]: row for row in csv.DictReader(f)}
k = stats["oc_jit_step_i32_v16"]
kj = {"cramped_room:65536": {"kernel": "oc_jit_step_i32_v16", "avg_us": float(k["AverageNs"]) / 1e3, "min_us": float(k["MinNs"]) / 1e3, "calls": int(k["Calls"]),
                             "measured": label,
                             "source": f"profiles/{tag}_kernel_stats.csv: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1000 --warmup 200 "
                                       "--no-cpu-baseline --no-secondary (every call the headline workload's)"}}
with open(os.path.join(P, "kernel_us.json"), "w") as f:
    json.dump(kj, f, indent=1)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(P, f"{tag}_pmc_fetch_size.csv"),
                os.path.join(P, f"{tag}_pmc_write_size.csv"), label], check=True, stdout=subprocess.DEVNULL)
with open(os.path.join(P, f"{tag}_kernel_stats_with_secondaries.csv")) as f:
    sec = {row["Name"]: row for row in csv.DictReader(f)}
pick = lambda frag: next(((n, r) for n, r in sec.items() if frag in n), (None, None))
line = json.loads([l for l in open(os.path.join(P, f"{tag}_bench_n1.log")) if l.startswith("{")][-1])
tj = json.load(open(os.path.join(P, "hbm_traffic.json")))["cramped_room:65536"]
print(f"N = 1 line: {line['value'] / 1e9:.2f} G agent-env steps/s, {line['roofline']['us_per_launch']:.2f} us per launch, roofline {line['roofline']['frac']:.3f}; "
      f"hbm_proper {line['roofline_hbm_proper']['us_per_launch']:.2f} us / {line['roofline_hbm_proper']['frac']:.3f}; policy_in_loop "
      f"{line['policy_in_loop']['us_per_step']:.1f} us per step (policy launch {line['policy_in_loop']['policy_kernel']['us_per_launch']:.1f}); "
      f"config 4's share {line['config4_share_policy_in_loop']['us_per_step']:.1f}; mixed play {line['mixed_play']['us_per_step']:.1f}")
print(f"rocprofv3: oc_jit_step_i32_v16 {k['Calls']} calls, average {float(k['AverageNs']):.0f} ns")
for frag in ("oc_jit_step_i32_v16_nt", "oc_jit_step_i32_v16_pl", "oc_policy_kernel", "oc_policy_duo_kernel", "np_group_kernel"):
    n, r = pick(frag)
    if n:
        print(f"  with secondaries: {n[:60]} {r['Calls']} calls, average {float(r['AverageNs']):.0f} ns")
print(f"traffic {tj['bytes_per_launch'] / 1e6:.2f} MB per launch (x{tj['bytes_per_launch'] / tj['algorithmic_bytes_per_launch']:.3f} of algorithmic)")
print("policy counters (sum over dispatches / last dispatch):", {c: (sum(v), v[-1]) for c, v in counters.items()})
