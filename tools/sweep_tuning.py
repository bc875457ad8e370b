#!/usr/bin/env python3
"""Launch-geometry sweep for the fused step kernel: kitchens-per-wave x staging-passes-per-flush.
Prints us/launch and achieved algorithmic GB/s per configuration (performance only; results are
identical for every setting -- tests/test_hip_parity.py::test_tuning_does_not_change_results)."""
import os as _os
_os.environ.setdefault("OC_DIAGNOSTICS", "1")      # phase stamps / ablation / experiment knobs live in csrc/liboc_hip_diag.so only
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--kitchens", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--kpw", default="8,16,32")
    ap.add_argument("--passes", default="1,2,4")
    ap.add_argument("--wpc", default="4,8,12,16", help="persistent waves per CU")
    args = ap.parse_args()
    import torch
    from bench import algorithmic_bytes
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params
    params = get_base_layout_params(args.layout, 200)
    S = params["width"] * params["height"]
    eng = OvercookedEngine(params, args.kitchens, 0)
    actions = torch.randint(0, 6, (64, 2, args.kitchens), device="cuda", dtype=torch.int32)
    nbytes = algorithmic_bytes(S) * args.kitchens
    print(f"# {args.layout} S={S} N={args.kitchens} bytes/launch={nbytes}")
    for kpw in [int(x) for x in args.kpw.split(",")]:
        for g in [int(x) for x in args.passes.split(",")]:
            for w in [int(x) for x in args.wpc.split(",")]:
                try:
                    eng.set_tuning(kpw, g, w)
                except Exception as e:
                    print(f"kpw={kpw:3d} passes={g} wpc={w:2d} skipped: {e}")
                    continue
                eng.rollout(actions, 50)
                best = min(eng.rollout_timed(actions, args.steps) for _ in range(3))
                us = best * 1e3 / args.steps
                print(f"kpw={kpw:3d} passes={g} wpc={w:2d}  {us:8.2f} us/launch  {nbytes / us / 1e3:8.1f} GB/s  frac={nbytes / us / 1e3 / 8000:.3f}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
