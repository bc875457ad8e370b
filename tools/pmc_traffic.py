#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md's HBM section prescribes):

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <round-label> [kernel] [grid]

Per launch of `kernel` with `grid` threads (default: the grid it was launched with most often -- run bench.py with
--no-secondary so that every launch of the kernel is the headline workload's): FETCH_SIZE x 2 (gfx950 counts a
wide coalesced read at half its bytes) + WRITE_SIZE (exact for 16-byte-per-lane stores), both reported in KB."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import algorithmic_bytes  # noqa: E402


def mean_counter(path, counter, kernel, grid):
    rows = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Kernel_Name"] == kernel and row["Counter_Name"] == counter:
                rows.append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    if not grid and rows:                               # the grid the kernel was launched with most often
        sizes = [g for g, _ in rows]
        grid = max(set(sizes), key=sizes.count)
    vals = [v for g, v in rows if g == grid]
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} with grid {grid} in {path}")
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_csv, write_csv, label = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "oc_jit_step_i32_v16"
    grid = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    fetch_kb, nf = mean_counter(fetch_csv, "FETCH_SIZE", kernel, grid)
    write_kb, nw = mean_counter(write_csv, "WRITE_SIZE", kernel, grid)
    total = int(round((2 * fetch_kb + write_kb) * 1024))
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data["cramped_room:65536"] = {
        "bytes_per_launch": total, "WRITE_SIZE_KB": write_kb, "FETCH_SIZE_KB_raw": fetch_kb,
        "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact",
        "algorithmic_bytes_per_launch": algorithmic_bytes(20) * 65536, "measured": label,
        "source": f"{os.path.basename(fetch_csv)} + {os.path.basename(write_csv)} under profiles/ (separate --pmc passes, {nf} / {nw} launches averaged)",
        "kernel": kernel}
    with open(path, "w") as f:
        json.dump(data, f, indent=1)
    print(json.dumps(data["cramped_room:65536"], indent=1))


if __name__ == "__main__":
    main()
