"""bench.py's multi-rank path executed for real: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both pinned to the one GPU of the test box
(OC_BENCH_DEVICE=0) with gloo as the control-plane backend (two NCCL ranks cannot share a device).  The rollout itself has
no data-path collective, so this exercises everything but the backend name: rendezvous, barriers, MAX over ranks, the JSON."""
import json
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_one_json_line(clean_launcher):
    one = clean_launcher.run([sys.executable, "bench.py", "--steps", "300", "--warmup", "50", "--no-cpu-baseline", "--no-policy"], timeout=500)
    assert one["rc"] == 0, one["stderr"][-2000:]
    lines1 = [ln for ln in one["stdout"].splitlines() if ln.startswith("{")]
    assert len(lines1) == 1
    j1 = json.loads(lines1[0])
    assert j1["n_gpus"] == 1 and j1["scaling"] == "weak" and j1["roofline"]["bound"] == "hbm"
    # the bare command form of the driver's N = 1 record with --gpus 2 and NO rendezvous in the environment: bench.py starts its two
    # ranks itself, as a child torch.distributed.run, relays rank 0's line and the exit code and never touches the GPU in the parent
    two = clean_launcher.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "300", "--warmup", "50"],
                             env={"OC_BENCH_DEVICE": "0", "OC_BENCH_BACKEND": "gloo", "OC_BENCH_EXCHANGE_T": "12"},
                             drop_env=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"), timeout=500)
    assert two["rc"] == 0, two["stderr"][-2000:]
    lines2 = [ln for ln in two["stdout"].splitlines() if ln.startswith("{")]
    assert len(lines2) == 1, "rank 0 prints exactly one JSON line"
    j2 = json.loads(lines2[0])
    assert j2["n_gpus"] == 2 and j2["steps"] == 300 and j2["warmup"] == 50 and j2["metric"] == j1["metric"]
    assert j2["launcher"] == "self" and j1["launcher"] == "none"
    assert j2["config"]["kitchens_per_gpu"] == 65536 and "x2" in j2["config"]["parallelism"]
    # whole-job throughput: twice the kitchens in (two ranks time-sharing one GPU) about twice the time; the value must be
    # computed from all ranks' kitchens and the slowest rank's time, so it lands near the single-rank figure, not at half or double
    assert 0.6 * j1["value"] < j2["value"] < 1.4 * j1["value"], (j1["value"], j2["value"])
    assert abs(j2["value"] - 2 * 65536 * 2 * 300 / (j2["ms_per_step"] * 300 / 1e3)) < 1e-3 * j2["value"]
    # the strong-scaling secondary: config 4's split is 65 536 / world kitchens per GPU
    assert j1["config4_split"]["kitchens_per_gpu"] == 65536 and j2["config4_split"]["kitchens_per_gpu"] == 32768
    assert "cpu_baseline" not in j2 and "hbm_proper" in j2
    # at N = 1 the exchange leg runs by default as a one-rank RCCL call on device buffers (full size: 65 536 kitchens x 200 steps)
    ex1 = j1["exchange"]
    assert ex1["checked"] is True and ex1["rccl"]["backend"] == "nccl" and ex1["rccl"]["world"] == 1 and ex1["rccl"]["version"] and ex1["steps"] == 200
    assert ex1["record_step_us"]["one_launch_with_record_destination"] < ex1["record_step_us"]["step_plus_export_states"]
    # round 6: the chunked, overlapped form of the same exchange is timed beside it (what the update waits for beyond the rollout) ...
    ov = ex1["overlap"]
    assert ex1["exposed_ms"] >= 0 and ov["collectives_per_update"] == ov["chunks"] + 1 == 9 and ov["rollout_with_overlapped_exchange_ms"] >= ov["rollout_alone_ms"] > 0
    assert ex1["update_amortised_value"] >= ex1["update_amortised_serial_value"] * 0.9
    # ... no hiprtc build ran inside any timed region (the step + export_states leg read 2978 us "per step" in round 5's driver record)
    assert j1["jit_builds_inside_timed_regions"] == 0 and j2["jit_builds_inside_timed_regions"] == 0 and j1["jit_builds"]["timed_regions"] >= 10
    assert ex1["record_step_us"]["step_plus_export_states"] < 100
    # ... the line separates host cost from kernel time and says where the in-place planes live
    assert j1["host_fixed_us"] > 0 and j1["value_kernel_time"] > j1["value"] and j1["wall_with_barrier_ms"] >= j1["ms_per_step"] * j1["steps"] * 0.99
    r = j1["roofline"]
    assert r["fits_infinity_cache"] is True and abs(r["destination_MB"] - 52.4) < 0.1 and 0.5 < r["frac_hbm_proper"] < r["frac"] < 1
    assert j1["fused_rollout"]["fits_infinity_cache"] is True and j1["config4_split"]["fits_infinity_cache"] is True
    assert set(j1["cold_start"]["oc_create_ms_8192_kitchens"]) == {"simple", "unident_s", "random1", "random0", "random3"}
    # ... and config 2 is in the line
    assert j1["config2"]["kitchens"] == 4096 and 0 < j1["config2"]["frac_of_8TBps"] < 1 and j1["config2"]["fused_rollout_us_per_step"] < j1["config2"]["us_per_launch"]
    # the exchange leg: config 4's compact trajectory through the one all-gather + featurize_all, both ranks' blocks checked
    # against the checksums their owners took (gloo moves host buffers here: the test hook says so in the record)
    ex = j2["exchange"]
    assert ex["checked"] is True and ex["steps"] == 12 and ex["kitchens_per_gpu"] == 32768 and ex["collectives_per_update"] == 1
    assert ex["rccl"]["world"] == 2 and ex["rccl"]["backend"] == "gloo" and [d["rank"] for d in ex["rccl"]["devices"]] == [0, 1]
    assert ex["exchange_ms"] > 0 and ex["refeaturize_ms"] > 0 and ex["GBps_per_gpu_sent"] > 0 and ex["update_amortised_value"] > 0
    assert abs(ex["MB_sent_per_gpu"] - (13 * 32768 * 208 + 12 * 32768 * (8 + 8 + 4)) / 1e6) < 0.01       # T+1 record slots, actions, reward [T,P,n], done
    # secondaries run their own fixed counts whatever --steps says
    assert j2["config4_split"]["steps"] == 1000 and j2["fused_rollout"]["steps"] == 1000 and j2["hbm_proper"]["steps"] == 960
    assert "roofline_hbm_proper" in j2 and j2["roofline_hbm_proper"]["bound"] == "hbm"


def test_bench_four_ranks_self_launched(clean_launcher):
    """More ranks than two: `python bench.py --gpus 4` (self-launched) with four gloo ranks pinned to the box's one GPU -- the GPU pool
    allows six processes on a card, the test session itself is one of them, so four ranks is what a 1-GPU box can host (the world-8
    offsets of the exchange are covered on the CPU: tests/test_sharding.py).  Checks the per-rank split of config 4, one device entry per
    rank, every rank's block of the exchange against its owner's checksums."""
    run = clean_launcher.run([sys.executable, "bench.py", "--gpus", "4", "--kitchens", "4096", "--steps", "50", "--warmup", "10", "--no-cpu-baseline",
                              "--no-policy"], env={"OC_BENCH_DEVICE": "0", "OC_BENCH_BACKEND": "gloo", "OC_BENCH_EXCHANGE_T": "4"},
                             drop_env=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"), timeout=900)
    assert run["rc"] == 0, run["stderr"][-2000:]
    lines = [ln for ln in run["stdout"].splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["launcher"] == "self" and j["config"]["kitchens_per_gpu"] == 4096 and "x4" in j["config"]["parallelism"]
    assert abs(j["value"] - 2 * 4096 * 4 * 50 / (j["ms_per_step"] * 50 / 1e3)) < 1e-3 * j["value"]
    assert j["config4_split"]["kitchens_per_gpu"] == 16384
    ex = j["exchange"]
    assert ex["checked"] is True and ex["steps"] == 4 and ex["kitchens_per_gpu"] == 16384 and ex["collectives_per_update"] == 1
    assert ex["rccl"]["world"] == 4 and [d["rank"] for d in ex["rccl"]["devices"]] == [0, 1, 2, 3]
    assert abs(ex["MB_received_per_gpu"] - 3 * ex["MB_sent_per_gpu"]) < 1e-6


def test_rccl_branch_runs_with_one_rank(clean_launcher):
    """The `nccl` (= RCCL) branches -- never taken by the gloo tests -- with a single rank on the box's one GPU: bench.py's
    process-group path, and the packed trajectory all-gather / score all-reduce of sharding.py as real RCCL calls."""
    run = lambda extra, env, port: clean_launcher.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", str(port)] + extra, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=400)
    b = run(["bench.py", "--gpus", "1", "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-policy"], {"OC_BENCH_FORCE_DIST": "1"}, 29543)
    assert b["rc"] == 0, b["stderr"][-2000:]
    j = json.loads([ln for ln in b["stdout"].splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["value"] > 1e9
    # the exchange leg as a real RCCL call on device buffers, full size (65 536 kitchens x 200 steps), checked
    ex = j["exchange"]
    assert ex["checked"] is True and ex["steps"] == 200 and ex["kitchens_per_gpu"] == 65536
    assert ex["rccl"]["backend"] == "nccl" and ex["rccl"]["world"] == 1 and ex["rccl"]["transport"] == "device buffers"
    assert len(ex["rccl"]["devices"]) == 1 and ex["rccl"]["devices"][0]["device_index"] == 0 and ex["rccl"]["version"]
    assert ex["exchange_ms"] > 0 and ex["refeaturize_ms"] > 0 and 0 < ex["update_amortised_value"] < j["config4_split"]["value"]
    r = run(["tests/_rccl_single_rank.py"], {}, 29545)
    assert r["rc"] == 0, r["stderr"][-2000:]
    out = json.loads([ln for ln in r["stdout"].splitlines() if ln.startswith("{")][0])
    assert out == {"ok": True, "backend": "nccl", "world": 1, "collectives": 2}
