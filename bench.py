#!/usr/bin/env python3
"""Headline benchmark: agent-env steps/s of the fused step+encode path (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]            # N=1; N>1 starts its own ranks as a child torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One rank per GPU, each owning `--kitchens` kitchens (weak scaling: kitchens are independent, the
rollout has no data-path collective).  A step = one launch of the fused kernel over every kitchen of
the rank with synthetic uniform-random joint actions already resident in HBM.  Prints ONE JSON line.

Secondary figures (never `value`) run with their OWN fixed step counts and warm-ups, whatever --steps says, so that a
short driver run and a long profiling run report the same things.  With more than one rank (or OC_BENCH_FORCE_DIST=1) the
`exchange` secondary times the one real collective of a sharded run: BASELINE config 4's per-GPU compact trajectory through
`TrajectoryExchange.gather_trajectory` (ONE all_gather_into_tensor: RCCL over xGMI) and `featurize_all`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
XGMI_LINK_GBS = 153.0        # one xGMI link, one direction (same guide); a GPU has 7, one per peer
SEC_STEPS, SEC_WARMUP = 1000, 100   # every secondary figure: its own fixed counts, independent of --steps / --warmup


def algorithmic_bytes(S, P=2):
    """bytes one kitchen-step must move.  obs i8 write + state read&write
    (4 B/tile + 6 B/player + 4 B timestep) + i32 actions read + i32 rewards write + i32 done write."""
    return P * S * (5 * P + 10) + 2 * (4 * S + 6 * P + 4) + 4 * P + 4 * P + 4


INFINITY_CACHE_MB = 256.0    # MI355X Infinity Cache (same guide): an in-place destination below this is absorbed on-die


def destination(mbytes):
    """What every in-place leg says about where its plane bytes go: a destination that fits the Infinity Cache is overwritten there
    step after step, so its 'achieved GB/s' is fabric throughput, not HBM bandwidth."""
    return {"destination_MB": round(mbytes, 1), "fits_infinity_cache": bool(mbytes <= INFINITY_CACHE_MB)}


class TimedRegions:
    """Every timed region of this file runs inside `with regions:` -- the library's hiprtc-build counter is read on both sides, and
    the JSON line carries the sum (`jit_builds_inside_timed_regions`, 0 or the line says so loudly): a compile takes about a
    second, a timed region is microseconds to milliseconds."""

    def __init__(self):
        from diverse_conventions_amd import native
        self._count = native.lib().oc_debug_jit_builds
        self.inside = 0
        self.regions = 0

    def __enter__(self):
        self._c0 = self._count()
        return self

    def __exit__(self, *exc):
        self.inside += self._count() - self._c0
        self.regions += 1
        return False

    def total_builds(self):
        return int(self._count())


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(params, kitchens, budget_s=10.0):
    """Oracle (CPU port of the reference algorithm) on the same workload, bounded sample.  The thread
    count is calibrated first (OpenMP over kitchens; oversubscribed boxes get slower with more threads)."""
    import numpy as np
    from oracle import oracle
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 6, size=(8, 2, kitchens)).astype(np.int32)

    def rate(threads, seconds, min_steps):
        env = oracle.OracleVecEnv(params, kitchens, nthreads=threads)
        env.step(acts[0])                                # warm-up (page-in, thread pool)
        t0 = time.perf_counter()
        steps = 0
        while True:
            env.step(acts[steps % 8])
            steps += 1
            el = time.perf_counter() - t0
            if (el > seconds and steps >= min_steps) or steps >= 200000:
                return 2 * kitchens * steps / el, steps, el

    cores = usable_cores()
    single, _, _ = rate(1, 1.5, 2)
    candidates = sorted({c for c in (2, 4, 8, 16, 32, 64, cores) if 1 < c <= cores})
    best_threads, best = 1, single
    for c in candidates:
        r, _, _ = rate(c, 0.7, 2)
        if r > best:
            best_threads, best = c, r
    value, steps, el = rate(best_threads, budget_s, 8)
    # SURVEY 8(d) asks for the C oracle on one core and on all cores at N = 4096 as well as at the headline's N
    small = None
    if kitchens != 4096:
        acts_big, kitchens_big = acts, kitchens
        kitchens = 4096
        acts = rng.integers(0, 6, size=(8, 2, kitchens)).astype(np.int32)
        s1, _, _ = rate(1, 1.0, 8)
        sb_threads, sb = 1, s1
        for c in candidates:
            r, _, _ = rate(c, 0.4, 8)
            if r > sb:
                sb_threads, sb = c, r
        sv, ssteps, sel = rate(sb_threads, 2.0, 64)
        small = {"kitchens": 4096, "value": sv, "cores": sb_threads, "value_1core": s1, "unit": "agent-env steps/s",
                 "sample": f"{ssteps} steps x 4096 kitchens ({sel:.1f} s)"}
        acts, kitchens = acts_big, kitchens_big
    # counterpart of BASELINE config 1: one kitchen, pure-Python restatement of the reference's step + encode
    from oracle.py_oracle import PyKitchen
    kitchen = PyKitchen(params)
    tape = rng.integers(0, 6, size=(4096, 2)).tolist()
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 2.0:
        for a in tape[:512]:
            kitchen.n_step(a)
        n1 += 512
    python_1env = 2 * n1 / (time.perf_counter() - t0)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": value, "unit": "agent-env steps/s", "cores": best_threads, "kind": "port", "cpu_model": cpu_model,
            "usable_cores": cores,
            "sample": f"{steps} steps x {kitchens} kitchens ({el:.1f} s), same layout and action law as the GPU run, "
                      f"C oracle with OpenMP over kitchens, best of {[1] + candidates} threads on a host with {cores} usable cores",
            "value_1core": single, "kitchens": kitchens, "at_4096_kitchens": small,
            "python_1env": {"value": python_1env, "cores": 1,
                            "note": "config 1 counterpart: pure-Python restatement (oracle/py_oracle.py) of the reference's "
                                    "SimplifiedOvercooked.n_step, 1 kitchen, same host"},
            "reference_python_survey_container": {"value": 64000, "cores": 1,
                                                  "note": "reference SimplifiedOvercooked.n_step, BASELINE.md section 2 (other host)"}}


def _checksum(t):
    """Position-sensitive checksum of an int8 tensor (numel a multiple of 4): (sum of words, sum of words x (1 + index mod 8191))
    as Python ints; temporaries are 2x the tensor."""
    import torch
    w = t.contiguous().view(-1).view(torch.int32).to(torch.int64)
    k = torch.arange(w.numel(), device=w.device, dtype=torch.int64).remainder_(8191).add_(1)
    return int(w.sum()), int((w * k).sum())


def check_distinct_devices(idents, allow_shared=False):
    """One rank per GPU is the contract of a multi-GPU run (one node): two ranks on the same device index end the run loudly --
    unless the OC_BENCH_DEVICE test hook pinned them there on purpose.  (The uuids travel in the record as evidence; the check
    itself does not lean on them: a runtime that reported one uuid for all GPUs must not take a correct 8-GPU run down.)"""
    keys = [d["device_index"] for d in idents]
    if len(set(keys)) != len(idents) and not allow_shared:
        raise SystemExit(f"exchange: {len(idents)} ranks but only {len(set(keys))} distinct devices ({idents}); one rank per GPU is the contract")
    return len(set(keys))


def exchange_leg(world, rank, local, backend, gen, c4_step_s, regions):
    """The one real collective of a sharded run (consumer: the reference's XP / MP losses read whole buffers,
    train/XD/xd.py:311-397): every rank's compact trajectory of BASELINE config 4 -- asymmetric_advantages, 65 536 / world
    kitchens, T = 200 steps of [state records | actions | reward | done], ONE flat int32 wire buffer that the rollout's own launches
    fill in place (TrajectoryBuffer(store_states=True): the step kernel writes the records, no export pass) -- handed AS IT IS to
    TrajectoryExchange.gather_buffer (ONE all_gather_into_tensor, no concatenation pass) and the receive side, featurize_all (one
    oc_encode_trajectory launch per peer).  Timed with events on the current stream after a full warm-up exchange; checked: every
    peer's re-featurised block against checksums its owner took of the observations its own engine wrote during the rollout.
    Returns the dict for the JSON line (same on every rank)."""
    import torch
    import torch.distributed as dist
    from diverse_conventions_amd import OvercookedEngine, TrajectoryBuffer, TrajectoryExchange, get_base_layout_params
    total = 65536
    T = int(os.environ.get("OC_BENCH_EXCHANGE_T", "200"))            # test hook: a shorter trajectory
    n = total // world
    params = get_base_layout_params("asymmetric_advantages", 200)
    eng = OvercookedEngine(params, n, local)
    dev = eng.device
    eng.stagger()                                                    # episode phases spread out: resets all along the trajectory
    traj = TrajectoryBuffer(eng, T, store_states=True, obs_slots=1)  # compact fields = views of one wire buffer; one obs slot, in place
    acts = torch.randint(0, 6, (T, 2, n), generator=gen, device=dev, dtype=torch.int32)
    probe_t = sorted({0, T // 2, T})
    own_sums = {}
    traj.start()
    for t in range(T):                                               # the rollout that fills the shard (not timed here): ONE launch per step
        if t in probe_t:
            own_sums[t] = _checksum(traj.obs[0])                     # what this rank's engine says slot t looks like
        traj.step(acts[t])
    own_sums[T] = _checksum(traj.obs[0])
    torch.cuda.synchronize()
    # what the record destination saves: the same T steps with an export_states launch behind each step (round 3's way)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scratch = torch.empty_like(traj.records[0])
    for t in range(3):                                               # warm-up of BOTH forms (round 5's line timed a hiprtc build here: the
        eng.step(acts[t])                                            # first in-place step after steps into time-major slots, 2978 us "per step")
        eng.export_states(out=scratch)
        eng.step(acts[t], records=scratch)
    torch.cuda.synchronize()
    with regions:
        e0.record()
        for t in range(T):
            eng.step(acts[t])
            eng.export_states(out=scratch)
        e1.record()
        torch.cuda.synchronize()
    two_pass_us = e0.elapsed_time(e1) * 1e3 / T
    with regions:
        e0.record()
        for t in range(T):
            eng.step(acts[t], records=scratch)
        e1.record()
        torch.cuda.synchronize()
    one_pass_us = e0.elapsed_time(e1) * 1e3 / T
    del scratch
    # control plane: who is where (fails loudly when two ranks share a device and that was not asked for)
    ident = {"rank": rank, "device_index": local, "name": torch.cuda.get_device_name(local)}
    try:
        ident["uuid"] = str(torch.cuda.get_device_properties(local).uuid)
    except Exception:                                                 # noqa: BLE001
        ident["uuid"] = None
    try:
        free, tot = torch.cuda.mem_get_info(local)
        ident["hbm_GB"] = round(tot / 1e9, 1)
    except Exception:                                                 # noqa: BLE001
        pass
    idents, sums_all = [None] * world, [None] * world
    if dist.is_initialized():
        dist.all_gather_object(idents, ident)
        dist.all_gather_object(sums_all, own_sums)
    else:
        idents, sums_all = [ident], [own_sums]
    check_distinct_devices(idents, allow_shared="OC_BENCH_DEVICE" in os.environ)
    # gloo (test hook OC_BENCH_BACKEND) moves host buffers; RCCL moves the device buffer as it is
    on_host = backend != "nccl"
    ex = TrajectoryExchange(always_collective=True)
    send = traj
    if on_host:
        from types import SimpleNamespace
        send = SimpleNamespace(T=traj.T, P=traj.P, engine=eng, wire=traj.wire.cpu())
    sent_bytes = 4 * traj.wire.numel()
    recv = torch.empty(world * traj.wire.numel(), dtype=torch.int32, device="cpu" if on_host else dev)     # persistent receive buffer
    out = torch.empty((T + 1, 2, world * n, eng.width, eng.height, 20), dtype=torch.int8, device=dev)

    def once():
        e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        got = ex.gather_buffer(send, out=recv)
        e1.record()
        if on_host:                                                   # events see nothing of a gloo transfer of host buffers
            wall = time.perf_counter() - t0
            records_all = got["records"].to(dev)
        else:
            records_all = got["records"]
        e2.record()
        ex.featurize_all(eng, records_all, out=out)
        e3.record()
        torch.cuda.synchronize()
        return got, (wall * 1e3 if on_host else e0.elapsed_time(e1)), e2.elapsed_time(e3)

    once()                                                            # warm-up: communicator set-up, allocations, code objects
    reps = 3
    with regions:
        runs = [once() for _ in range(reps)]
    got = runs[-1][0]
    gather_ms = sum(r[1] for r in runs) / reps
    feat_ms = sum(r[2] for r in runs) / reps
    # what round 3's path paid on top: one concatenation of the four loose fields into a send buffer (read + write of the shard)
    cat_ms = None
    if not on_host:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        loose = (traj.records.reshape(-1), traj.actions.reshape(-1), traj.reward.reshape(-1), traj.done.reshape(-1))
        torch.cat(loose)
        e0.record()
        torch.cat(loose)
        e1.record()
        torch.cuda.synchronize()
        cat_ms = e0.elapsed_time(e1)
    # check: own fields come back at index `rank`, and every peer's block of the re-featurised tensor has the checksum its owner
    # computed from its own engine's observation of that slot
    own = {"records": send.wire[:traj.records.numel()].view(traj.records.shape)}
    ok = bool(torch.equal(got["records"][rank], own["records"]))
    ok = ok and all(torch.equal(got[k][rank].to(dev), getattr(traj, k)) for k in ("actions", "reward", "done"))
    for r in range(world):
        for t in probe_t:
            ok = ok and _checksum(out[t, :, r * n:(r + 1) * n]) == tuple(sums_all[r][t])
    # (after the checks of the one-shot exchange above: what follows rolls the trajectory out again)
    # ---- the same update with the exchange CHUNKED and OVERLAPPED with the rollout (ChunkedGather): the rollout's
    # launches fill the wire buffer chunk by chunk on the current stream; behind each chunk's last step a side stream all-gathers the
    # chunk's record slots and re-featurises them.  exposed = (rollout with the overlapped exchange) - (the same rollout alone).
    chunk = max(1, int(os.environ.get("OC_BENCH_EXCHANGE_CHUNK", "25")))
    n_chunks = (T + chunk - 1) // chunk
    ex2 = TrajectoryExchange(always_collective=True)
    out2 = torch.empty_like(out) if not on_host else None

    def rollout_in_chunks(cg):
        traj.start()
        for c in range(n_chunks):
            a0, a1 = c * chunk, min((c + 1) * chunk, T)
            eng.rollout(acts[a0:a1], a1 - a0, obs=traj.obs[0], reward=traj.reward[a0:a1], done=traj.done[a0:a1], records=traj.records[a0 + 1:a1 + 1])
            if cg is not None:
                cg.submit(c)
        return cg.finish() if cg is not None else None

    exposed_ms = overlap_total_ms = rollout_alone_ms = None
    chunk_ok = True
    if not on_host:
        cg = ex2.chunked(traj, chunk, engine=eng, out=out2)
        rollout_in_chunks(None)
        res2 = rollout_in_chunks(cg)                                  # warm-up of both, and the check below
        torch.cuda.synchronize()
        chunk_ok = bool(torch.equal(cg.records_all()[rank], traj.records)) and all(
            torch.equal(res2[k][rank], getattr(traj, k)) for k in ("actions", "reward", "done"))
        for r in range(world):                                        # (the rollout restarted from the engine's current states: own blocks only)
            if r == rank:
                chunk_ok = chunk_ok and _checksum(out2[T, :, r * n:(r + 1) * n]) == _checksum(traj.obs[0])
        alone, both = [], []
        for _ in range(reps):
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()
            with regions:
                t0 = time.perf_counter()
                rollout_in_chunks(None)
                torch.cuda.synchronize()
                alone.append(time.perf_counter() - t0)
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()
            with regions:
                t0 = time.perf_counter()
                rollout_in_chunks(cg)
                torch.cuda.synchronize()
                both.append(time.perf_counter() - t0)
        rollout_alone_ms, overlap_total_ms = min(alone) * 1e3, min(both) * 1e3
        exposed_ms = max(0.0, overlap_total_ms - rollout_alone_ms)
        del out2
    stats = torch.tensor([gather_ms, feat_ms, 0.0 if (ok and chunk_ok) else 1.0, exposed_ms or 0.0, overlap_total_ms or 0.0, rollout_alone_ms or 0.0],
                         dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist.is_initialized():
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    gather_ms, feat_ms, bad, exposed_max, overlap_max, alone_max = (float(x) for x in stats)
    if exposed_ms is not None:
        exposed_ms, overlap_total_ms, rollout_alone_ms = exposed_max, overlap_max, alone_max
    eng.close()
    del out, recv
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                                 # noqa: BLE001
        rccl_version = None
    serial_update_s = T * c4_step_s + gather_ms / 1e3 + feat_ms / 1e3
    update_s = T * c4_step_s + (exposed_ms / 1e3 if exposed_ms is not None else gather_ms / 1e3 + feat_ms / 1e3)
    return {
        "what": f"BASELINE config 4's per-update exchange: asymmetric_advantages, {total} kitchens in total = {n} per GPU, T = {T} compact "
                "steps (T+1 state record slots + actions + reward + done: one flat wire buffer the step launches fill in place) through ONE "
                "all_gather_into_tensor as it is, then every peer's shard re-featurised on the receiver (oc_encode_trajectory, one launch per peer)",
        "steps": T, "kitchens_per_gpu": n, "collectives_per_update": 1, "reps": reps,
        "MB_sent_per_gpu": sent_bytes / 1e6, "MB_received_per_gpu": sent_bytes * (world - 1) / 1e6,
        "exchange_ms": gather_ms, "GBps_per_gpu_sent": sent_bytes * max(world - 1, 1) / gather_ms / 1e6,
        "exposed_ms": exposed_ms,
        "overlap": {"chunk_steps": chunk, "chunks": n_chunks, "collectives_per_update": n_chunks + 1,
                    "rollout_alone_ms": rollout_alone_ms, "rollout_with_overlapped_exchange_ms": overlap_total_ms,
                    "serial_exchange_plus_refeaturize_ms": gather_ms + feat_ms,
                    "note": "ChunkedGather: behind the last step of every chunk of the rollout a side stream all-gathers the chunk's record "
                            "slots and re-featurises every peer's block; exposed_ms = what the update waits for beyond the rollout itself "
                            "(best of the repetitions; at one rank on one GPU both the rollout and the re-featurisation are HBM-bound, so little "
                            "can hide -- on 8 GPUs the rollout is 8192 kitchens with the policies in the loop, matrix-bound, DESIGN.md 7)"},
        "concatenation_ms_saved": cat_ms,
        "record_step_us": {"one_launch_with_record_destination": one_pass_us, "step_plus_export_states": two_pass_us},
        "model_ms_direct_all_gather": sent_bytes / XGMI_LINK_GBS / 1e6,
        "model_note": f"direct all-gather: a GPU pushes its shard to each peer over that peer's own xGMI link ({XGMI_LINK_GBS:.0f} GB/s each, "
                      "all links at once), so the model time is shard bytes / one link whatever the world size; a single ring would take (world-1) x that",
        "refeaturize_ms": feat_ms, "refeaturize_GB_written": (T + 1) * 2 * world * n * eng.size * 20 / 1e9,
        "refeaturize_GBps": (T + 1) * 2 * world * n * eng.size * 20 / feat_ms / 1e6,
        "checked": bad == 0.0,
        "rccl": {"backend": backend if dist.is_initialized() else None, "world": world, "version": rccl_version,
                 "transport": "device buffers" if not on_host else "host buffers (gloo test hook)", "devices": idents},
        "update_amortised_value": 2 * total * T / update_s, "update_amortised_unit": "agent-env steps/s",
        "update_amortised_serial_value": 2 * total * T / serial_update_s,
        "update_amortised_note": f"config 4 over one PPO update: {T} env steps at the config4_split step time + the EXPOSED part of the chunked, "
                                 "overlapped exchange and re-featurisation (update_amortised_serial_value: + the whole one-shot exchange and "
                                 "re-featurisation, rounds 3-5's figure), whole job",
    }


def self_launch(n_ranks):
    """Start the N ranks of `bench.py --gpus N` the way the driver does -- python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments> -- as a child process, relay its
    output and return its exit code.  The caller (this process) never touches the GPU: on this pool a process that has initialised
    the GPU must not start other GPU programs, and the parent has no use for a device anyway."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, OC_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)      # stderr goes straight through
    lines = 0
    for line in proc.stdout:
        if line.startswith("{"):
            lines += 1
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py launcher: expected ONE JSON line from rank 0, saw {lines}", file=sys.stderr)
        return 1
    return rc


def check_policy_in_loop(ro, eng, params, T=24, blocks=4, per_block=64):
    """Outside every timed region: T more steps of the policies-in-the-loop rollout recorded into a trajectory, then `blocks` x
    `per_block` kitchens (spread over the pairing segments) replayed through the CPU oracle from their exported start states
    with the recorded actions -- observations, rewards and dones must be bit-identical."""
    import numpy as np
    import torch
    from diverse_conventions_amd import TrajectoryBuffer
    from oracle import oracle
    n = eng.num_kitchens
    firsts = [int(i * (n - per_block) / max(blocks - 1, 1)) for i in range(blocks)]
    starts = [eng.get_state(f, per_block) for f in firsts]
    traj = TrajectoryBuffer(eng, T)
    ro.collect(traj)
    torch.cuda.synchronize()
    for f, (tiles, players, ts) in zip(firsts, starts):
        ora = oracle.OracleVecEnv(params, per_block)
        ora.set_state(tiles, players, ts)
        sl = slice(f, f + per_block)
        if not np.array_equal(traj.obs[0][:, sl].cpu().numpy(), ora.encode()):
            return False
        for t in range(T):
            a = traj.actions[t][:, sl].cpu().numpy()
            if a.min() < 0 or a.max() > 5:
                return False
            oobs, orew, odone = ora.step(np.ascontiguousarray(a))
            if not (np.array_equal(traj.obs[t + 1][:, sl].cpu().numpy(), oobs) and np.array_equal(traj.reward[t][:, sl].cpu().numpy(), orew)
                    and np.array_equal(traj.done[t][sl].cpu().numpy(), odone)):
                return False
    return True


def drop_in_and_eval_legs(local, regions, gen):
    """Two secondaries about the surface the north star promises: the reference's training loop calls
    `env.step(ego_action)` once per step from Python (pantheonrl_extension/vectorenv.py:171-200, train/MAPPO/main_player.py:211-277),
    not a C-enqueued run of launches.
      drop_in_step: microseconds per OvercookedMadrona.step() -- ego actions already on the GPU -- with a RandomVectorAgent partner and
                    with an eager-PyTorch R_Actor partner (the reference's network, utils/cnn.py:26-38), at 65 536 and 8192 kitchens,
                    beside the env kernel's own time and the host's share (the loop's enqueue time before the final synchronize);
      eval_1000:    the authors' evaluation shape (train/testing.py:39-59): 1000 kitchens x 200 steps, two decentralised policies
                    (here both through FusedVectorAgent), reward / 20 accumulated; the recorded actions replayed through the CPU
                    oracle afterwards (rewards and dones bit-identical)."""
    import numpy as np
    import torch
    from diverse_conventions_amd import (FusedVectorAgent, OvercookedMadrona, PackedNetworks, RandomVectorAgent, VectorAgent, VectorMultiAgentEnv,
                                         get_base_layout_params)
    from oracle import oracle
    W, H = 5, 4
    P = (W - 2) * (H - 2)
    g0 = torch.Generator().manual_seed(7)
    r = lambda *sh: torch.randn(*sh, generator=g0) * 0.1
    sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
            "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
            "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(2)]

    class TorchActor(VectorAgent):
        """the reference's R_Actor as plain eager PyTorch: CNNLayer (conv 3x3 -> ReLU -> flatten -> Linear -> ReLU -> Linear -> ReLU) +
        Categorical head, sampled (train/MAPPO/utils/cnn.py:26-41, utils/act.py:41-80).  The 3x3 valid convolution is applied as one
        dense matrix over the flattened (W, H, C) row -- the same linear map, without a MIOpen kernel search on a fresh box."""

        def __init__(self, sd, dev):
            cw = sd["base.cnn.cnn.0.weight"]                             # [O, C, 3, 3], multiplies input tile (x + kx, y + ky)
            dense = torch.zeros((W, H, 20, 32, W - 2, H - 2))            # -> output (o, x', y'): the flatten order of the reference
            for x in range(W - 2):
                for y in range(H - 2):
                    dense[x:x + 3, y:y + 3, :, :, x, y] = cw.permute(2, 3, 1, 0)
            self.w0 = dense.reshape(W * H * 20, 32 * P).to(dev)
            self.b0 = sd["base.cnn.cnn.0.bias"].repeat_interleave(P).to(dev)
            self.w1, self.b1 = sd["base.cnn.cnn.3.weight"].t().contiguous().to(dev), sd["base.cnn.cnn.3.bias"].to(dev)
            self.w2, self.b2 = sd["base.cnn.cnn.5.weight"].t().contiguous().to(dev), sd["base.cnn.cnn.5.bias"].to(dev)
            self.w3, self.b3 = sd["act.action_out.linear.weight"].t().contiguous().to(dev), sd["act.action_out.linear.bias"].to(dev)

        def logits(self, obs):
            x = obs.reshape(obs.shape[0], -1).float()
            x = torch.relu(torch.addmm(self.b0, x, self.w0))
            x = torch.relu(torch.addmm(self.b1, x, self.w1))
            x = torch.relu(torch.addmm(self.b2, x, self.w2))
            return torch.addmm(self.b3, x, self.w3)

        @torch.no_grad()
        def get_action(self, obs, record=True):
            return torch.distributions.Categorical(logits=self.logits(obs.obs)).sample().unsqueeze(-1)

        def update(self, rewards, dones):
            pass

    drop = {"what": "microseconds per OvercookedMadrona.step(ego_action) called from Python as the reference's runner does "
                    "(vectorenv.py:171-200): partner.get_action, the [P, N, 1] action assembly, ONE fused env launch, the ego view; ego "
                    "actions resident on the GPU; cramped_room", "cases": []}
    T, t_act = 400, 64
    for n in (65536, 8192):
        env = OvercookedMadrona("cramped_room", n, local, debug_compile=False)
        dev = env.sim.device
        ego = torch.randint(0, 6, (t_act, n, 1), generator=gen, device=dev, dtype=torch.int64)
        par = torch.randint(0, 6, (t_act, n, 1), generator=gen, device=dev, dtype=torch.int64)
        # the env kernel alone at this size, C-enqueued
        flat = torch.randint(0, 6, (t_act, 2, n), generator=gen, device=dev, dtype=torch.int32)
        env.sim.rollout(flat, 50)
        with regions:
            kern_us = env.sim.rollout_timed(flat, 500) * 1e3 / 500
        case = {"kitchens": n, "env_kernel_us": kern_us}
        for name in ("random_partner", "random_partner_stacked", "random_partner_graph_step", "torch_actor_partner"):
            tick = [0]
            agent = TorchActor(sds[1], dev) if name == "torch_actor_partner" else RandomVectorAgent(lambda: par[tick[0] % t_act])
            env.partners[0].clear()
            env.add_partner_agent(agent)
            env._graph_step = name == "random_partner_graph_step"   # n_step replays a one-launch HIP graph of the env step instead of enqueueing it
            # (_stacked: the reference's own VectorMultiAgentEnv.step -- torch.stack into [P, N, 1], then n_step -- instead of the class's
            # two-buffer step, oc_step_players)
            step_fn = (lambda act: VectorMultiAgentEnv.step(env, act)) if name == "random_partner_stacked" else env.step
            env.reset()
            for t in range(60):
                tick[0] = t
                step_fn(ego[t % t_act])
            torch.cuda.synchronize()
            best = None
            for _ in range(2):
                with regions:
                    t0 = time.perf_counter()
                    for t in range(T):
                        tick[0] = t
                        step_fn(ego[t % t_act])
                    t1 = time.perf_counter()
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                cur = {"us_per_step": (t2 - t0) / T * 1e6, "host_enqueue_us_per_step": (t1 - t0) / T * 1e6}
                best = cur if best is None or cur["us_per_step"] < best["us_per_step"] else best
            best["x_env_kernel"] = best["us_per_step"] / kern_us
            best["value"] = 2 * n / best["us_per_step"] * 1e6
            case[name] = best
        env.close()
        drop["cases"].append(case)
    drop["note"] = ("host-bound when host_enqueue_us_per_step ~ us_per_step: the Python of one step costs more than the kernel.  The class's own "
                    "step hands the ego's and the partner's action tensors to the kernel as they are (oc_step_players, round 6); `_stacked` is the "
                    "reference's way (partner call, torch.stack into the persistent [P, N, 1] buffer, n_step); `_graph_step` replays the env "
                    "launch as a one-kernel HIP graph (OvercookedMadrona(graph_step=True)) -- a graph of one kernel saves nothing, the cost is "
                    "the Python around it; the C-enqueued loops of this line (`value`, policy_in_loop) are what a runner that hands the loop "
                    "to the library gets")

    # ---- eval_1000: train/testing.py:39-59
    n, L = 1000, 200
    params = get_base_layout_params("cramped_room", L)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
    from diverse_conventions_amd import FusedAgentPair

    def episode_leg(make_agents):
        """The loop as the reference writes it, with the two agents `make_agents` returns; timed over one episode, then a recorded
        episode replayed through the CPU oracle."""
        env = OvercookedMadrona("cramped_room", n, local, debug_compile=False, horizon=L)
        dev = env.sim.device
        ego_agent, alt_agent = make_agents()
        env.add_partner_agent(alt_agent)
        rec_actions = torch.zeros((L, 2, n), dtype=torch.int32, device=dev)
        rec_rew = torch.zeros((L, n), dtype=torch.int32, device=dev)
        rec_done = torch.zeros((L, n), dtype=torch.int32, device=dev)

        def run_sim(record):
            obs = env.reset()
            reward = torch.zeros(n, device=dev)
            for i in range(L):
                action = ego_agent.get_action(obs, False)
                obs, newreward, done, _ = env.step(action)
                reward += newreward / 20
                if record:
                    rec_actions[i, 0].copy_(ego_agent.actions)   # (the agents keep their last actions; the env read both buffers as they were)
                    rec_actions[i, 1].copy_(alt_agent.actions)
                    rec_rew[i].copy_(newreward)
                    rec_done[i].copy_(done)
            return reward

        env.sim.reset_all()
        run_sim(False)                                               # warm-up episode (ends exactly at the horizon: every kitchen resets)
        torch.cuda.synchronize()
        with regions:
            t0 = time.perf_counter()
            reward = run_sim(False)
            torch.cuda.synchronize()
            ev_s = time.perf_counter() - t0
        start = env.sim.get_state(0, n)                              # (the timed episode ended at the horizon too: a fresh episode starts here)
        reward = run_sim(True)
        torch.cuda.synchronize()
        ora = oracle.OracleVecEnv(params, n)
        ora.set_state(*start)
        ok = True
        a_all, r_all, d_all = rec_actions.cpu().numpy(), rec_rew.cpu().numpy(), rec_done.cpu().numpy()
        total = np.zeros(n)
        for i in range(L):
            _, orew, odone = ora.step(np.ascontiguousarray(a_all[i]))
            ok = ok and np.array_equal(orew[0], r_all[i]) and np.array_equal(np.asarray(odone).astype(bool), d_all[i].astype(bool))
            total += orew[0] / 20
        ok = ok and np.allclose(total, reward.cpu().numpy())
        env.close()
        return ev_s, float(reward.mean()), bool(ok)

    ev_s, mean_ret, ok = episode_leg(lambda: (FusedVectorAgent(nets, n, [(0, n, 0)], seed=11), FusedVectorAgent(nets, n, [(0, n, 1)], seed=12)))
    pair_box = []

    def make_pair():
        pair_box.append(FusedAgentPair(nets, n, [(0, n, 0)], [(0, n, 1)], seed=11))
        return pair_box[0].ego, pair_box[0].partner
    pair_s, pair_ret, pair_ok = episode_leg(make_pair)
    pair_ok = pair_ok and pair_box[0].single_launches == 0 and pair_box[0].joint_launches == 3 * L
    # the same evaluation with the loop handed to the library: FusedPairedRollout over ONE segment (ego = network 0, partner = network 1),
    # the episode's 200 steps -- one policy launch + one env launch each -- enqueued by one call into a trajectory, reward / 20 summed once
    from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, TrajectoryBuffer
    eng_l = OvercookedEngine(params, n, local)
    ro_l = FusedPairedRollout(eng_l, nets, n, 0, actor_nets=[[0], [1]], seed=11)
    traj_l = TrajectoryBuffer(eng_l, L)
    ro_l.collect(traj_l)                                             # warm-up episode (ends at the horizon: every kitchen resets)
    traj_l.after_update()
    torch.cuda.synchronize()
    with regions:
        t0 = time.perf_counter()
        ro_l.collect(traj_l)
        lib_ret = traj_l.reward[:, 0].sum(0) / 20
        torch.cuda.synchronize()
        lib_s = time.perf_counter() - t0
    traj_l.after_update()
    lib_ok = check_policy_in_loop(ro_l, eng_l, params, blocks=3, per_block=48)
    lib_mean = float(lib_ret.float().mean())
    eng_l.close()
    ev = {"what": "the reference's evaluation loop (train/testing.py:39-59): cramped_room, 1000 kitchens x 200 steps, two decentralised "
                  "policies (ego and partner each one FusedVectorAgent launch per step), env.step from Python, reward / 20 accumulated",
          "kitchens": n, "steps": L, "episode_ms": ev_s * 1e3, "us_per_step": ev_s / L * 1e6, "value": 2 * n * L / ev_s, "unit": "agent-env steps/s",
          "launches_per_step": 3, "mean_return_over_20": mean_ret, "checked": bool(ok),
          "checked_note": "a third episode recorded (joint actions, rewards, dones) and replayed through the CPU oracle from the exported start "
                          "state: rewards and dones bit-identical at every step, the accumulated reward / 20 equal",
          "agent_pair": {"what": "the same loop, unchanged, with the two agents of a FusedAgentPair: the ego's get_action evaluates both players' "
                                 "rows of the simulator's [2, N, ...] observation tensor in ONE policy launch and the partner's, called inside "
                                 "env.step, hands its half out (round 6)",
                         "episode_ms": pair_s * 1e3, "us_per_step": pair_s / L * 1e6, "value": 2 * n * L / pair_s, "launches_per_step": 2,
                         "mean_return_over_20": pair_ret, "checked": bool(pair_ok)},
          "library_loop": {"what": "the same evaluation with the loop handed to the library (FusedPairedRollout, one segment: ego network 0, "
                                   "partner network 1): 200 x (one policy launch + one env launch) enqueued by ONE call into a time-major "
                                   "trajectory, reward / 20 summed at the end",
                           "episode_ms": lib_s * 1e3, "us_per_step": lib_s / L * 1e6, "value": 2 * n * L / lib_s, "launches_per_step": 2,
                           "mean_return_over_20": lib_mean, "checked": bool(lib_ok)},
          "note": "launch-bound: 1000 kitchens are 2 MB of planes per step; a policy launch is its fixed part (the network into LDS, one 32-row "
                  "block per workgroup) whatever the rows"}
    return drop, ev



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--kitchens", type=int, default=65536, help="kitchens per GPU")
    ap.add_argument("--layout", default="cramped_room")
    ap.add_argument("--kpw", type=int, default=0, help="kitchens per wave (0 = engine default)")
    ap.add_argument("--passes", type=int, default=0, help="staging passes per flush (0 = engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-policy", action="store_true", help="skip the policies-in-the-loop secondary figure")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline measurement only (profiling runs: every launch of the step kernel is then the headline workload's)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no rendezvous in the environment: this process becomes the launcher.  It has made
    # no GPU call (torch is not even imported yet) and makes none: the N ranks are CHILD processes of torch.distributed.run, whose
    # one JSON line and exit code are relayed.  Under an external launcher (WORLD_SIZE set) nothing of this runs.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # stdout carries the ONE JSON line and nothing else: libraries that print banners there (RCCL's version block on communicator
    # set-up) are sent to stderr by pointing fd 1 at fd 2 for the life of the process; the line goes out through a copy of the real fd
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from diverse_conventions_amd import OvercookedEngine, get_base_layout_params

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    launcher = os.environ.get("OC_BENCH_LAUNCHER", "external" if "WORLD_SIZE" in os.environ else "none")
    # test hooks (a 1-GPU box cannot host two NCCL ranks): OC_BENCH_DEVICE pins every rank to one device,
    # OC_BENCH_BACKEND=gloo swaps the control-plane backend; the driver's multi-GPU runs use neither
    if "OC_BENCH_DEVICE" in os.environ:
        local = int(os.environ["OC_BENCH_DEVICE"])
    backend = os.environ.get("OC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    # OC_BENCH_FORCE_DIST=1: run the process-group code (RCCL init, barriers, the MAX all-reduce) even with one rank, so
    # that a 1-GPU box executes the nccl branch too
    use_dist = world > 1 or bool(os.environ.get("OC_BENCH_FORCE_DIST"))
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    params = get_base_layout_params(args.layout, 200)
    S = params["width"] * params["height"]
    n = args.kitchens
    torch.cuda.synchronize()                             # (the HIP context exists: what follows is the engine's own set-up)
    t_create = time.perf_counter()
    eng = OvercookedEngine(params, n, local)
    headline_create_ms = (time.perf_counter() - t_create) * 1e3
    if args.kpw or args.passes:
        eng.set_tuning(args.kpw, args.passes)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    t_act = 64
    actions = torch.randint(0, 6, (t_act, 2, n), generator=gen, device="cuda", dtype=torch.int32)

    # the bracket barrier: an all-reduce of one pre-allocated element (nobody's completes before everybody has entered) followed by
    # the device sync -- 20 us with one RCCL rank where dist.barrier(), which allocates its own tensor, takes 28
    # (tools/exp_barrier_cost.py); a 20-step timed region is 270 us
    bar_token = torch.zeros(1, device="cuda" if backend == "nccl" else "cpu") if use_dist else None

    def barrier():
        if use_dist:
            dist.all_reduce(bar_token)
        torch.cuda.synchronize()

    # W untimed warm-up steps; the last of them goes through the timed entry point, so that the timed region is not the first
    # caller of anything (the runtime sets up its event machinery on first use: tens of microseconds of a 20-step run)
    if args.warmup > 1:
        eng.rollout(actions, args.warmup - 1)
    if args.warmup > 0:
        eng.rollout_timed(actions, 1)
    regions = TimedRegions()
    builds_before_timing = regions.total_builds()        # hiprtc builds of oc_create + warm-up (0 when the ahead-of-time objects match)
    barrier()
    with regions:
        t0 = time.perf_counter()
        kernel_ms = eng.rollout_timed(actions, args.steps)   # hipEvents on the launch stream bracket the K launches
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0                  # this rank's K steps: from behind the opening barrier + sync to its own sync; MAX over ranks below
        barrier()                                        # (the closing barrier: behind the clock -- with N ranks it is an all-reduce whose own latency,
        wall_with_barrier = time.perf_counter() - t0     #  tens of microseconds, is not part of the K steps a 20-step run times; the barrier-to-barrier
                                                         #  window of rounds 1-4 travels beside it as `wall_with_barrier_ms`)
    create_ms = {}
    fused_ms = tm_us = c4_wall = c4_kernel_ms = float("nan")
    slots, c4_total, c4_n, c4_S, c4_steps, c3 = 64, 65536, 65536 // world, 45, SEC_STEPS, None
    if not args.no_secondary:
        # Secondary figure, never `value`: SEC_STEPS steps as ONE launch (oc_rollout_fused: per-kitchen state stays in
        # LDS between steps, same action tape read from HBM every step, every step's obs/reward/done written).
        # Only usable when actions are known ahead, which is exactly this synthetic rollout; a policy in the loop
        # needs the one-launch-per-step path that `value` measures.
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.rollout_fused(SEC_WARMUP, actions=actions)
        with regions:
            ev0.record()
            eng.rollout_fused(SEC_STEPS, actions=actions)
            ev1.record()
            torch.cuda.synchronize()
        fused_ms = ev0.elapsed_time(ev1)
        # Secondary figure, never `value`: HBM proper.  The same launches with every step's observations / rewards / dones
        # written to their own time-major trajectory slot (what a rollout buffer wants): 64 slots x 52 MB are
        # far beyond the 256 MB Infinity Cache, so no byte is absorbed on-die (the in-place buffer of `value` is the
        # reference's data-plane contract -- the simulator's persistent output tensors, overwritten every step -- and fits it).
        barrier()
        slots = 64
        tm_obs = torch.empty((slots,) + tuple(eng.obs.shape), dtype=torch.int8, device="cuda")
        tm_rew = torch.empty((slots, 2, n), dtype=torch.int32, device="cuda")
        tm_done = torch.empty((slots, n), dtype=torch.int32, device="cuda")
        for _ in range(2):
            eng.rollout(actions, slots, obs=tm_obs, reward=tm_rew, done=tm_done)
        tm_rounds = max(1, SEC_STEPS // slots)
        with regions:
            ev0.record()
            for _ in range(tm_rounds):
                eng.rollout(actions, slots, obs=tm_obs, reward=tm_rew, done=tm_done)
            ev1.record()
            torch.cuda.synchronize()
        tm_us = ev0.elapsed_time(ev1) * 1e3 / (tm_rounds * slots)
        del tm_obs, tm_rew, tm_done

        # Secondary figure, never `value`: BASELINE config 4's split -- asymmetric_advantages, 65 536 kitchens in total, i.e.
        # 65 536 / N per GPU (strong scaling; 8192 per GPU on the 8-GPU node the config names).
        barrier()
        c4_total = 65536
        c4_n = c4_total // world
        c4_params = get_base_layout_params("asymmetric_advantages", 200)
        c4_S = c4_params["width"] * c4_params["height"]
        eng4 = OvercookedEngine(c4_params, c4_n, local)
        act4 = torch.randint(0, 6, (t_act, 2, c4_n), generator=gen, device="cuda", dtype=torch.int32)
        c4_steps = SEC_STEPS
        eng4.rollout(act4, SEC_WARMUP)
        barrier()
        with regions:
            t4 = time.perf_counter()
            c4_kernel_ms = eng4.rollout_timed(act4, c4_steps)
            torch.cuda.synchronize()
            c4_wall = time.perf_counter() - t4
        barrier()
        eng4.close()

        # Secondary figure, never `value`: BASELINE config 3's env step -- the five classic layouts x 8192 kitchens stepped by
        # ONE launch (oc_group_step)
        c3 = None
        if world == 1:
            from diverse_conventions_amd import CLASSIC_LAYOUTS, EngineGroup
            # cold start: oc_create of the five classic layouts at config 3's size, the first engines of these shapes in this process
            # (module from the ahead-of-time objects, state block, tables, start-state fill) -- VERDICT r05 item 7 asks for < 50 ms each
            for name in CLASSIC_LAYOUTS:
                tc = time.perf_counter()
                e_cs = OvercookedEngine(get_base_layout_params(name, 200), 8192, local)
                create_ms[name] = {"ms": (time.perf_counter() - tc) * 1e3, "kernel_build": e_cs.kernel_build_info.split("; ")[-1].rstrip(")")}
                e_cs.close()
            grp = EngineGroup.from_layouts(CLASSIC_LAYOUTS, 8192, local)
            act3 = [torch.randint(0, 6, (t_act, 2, 8192), generator=gen, device="cuda", dtype=torch.int32) for _ in grp.engines]
            grp.rollout_timed(act3, SEC_WARMUP)
            c3_steps = SEC_STEPS
            with regions:
                c3_us = grp.rollout_timed(act3, c3_steps) * 1e3 / c3_steps
            c3_bytes = sum(algorithmic_bytes(e.size) * e.num_kitchens for e in grp.engines)
            c3 = {"what": "BASELINE config 3 env step: 5 classic layouts x 8192 kitchens, ONE launch per joint step (oc_group_step)",
                  "steps": c3_steps, "us_per_joint_step": c3_us, "value": 2 * 5 * 8192 / c3_us * 1e6, "unit": "agent-env steps/s",
                  "bytes_per_launch": c3_bytes, "achieved_GBps": c3_bytes / c3_us / 1e3, "frac_of_8TBps": c3_bytes / c3_us / 1e3 / HBM_PEAK_GBS,
                  **destination(sum(2 * e.num_kitchens * e.size * 20 for e in grp.engines) / 1e6), "build": grp.build_info}
            grp.close(close_engines=True)

    # every rank's timings -> the slowest rank's (MAX), as the contract asks for `value`
    stats = torch.tensor([wall, kernel_ms, fused_ms, tm_us, c4_wall, c4_kernel_ms, wall_with_barrier], dtype=torch.float64,
                         device="cuda" if backend == "nccl" else "cpu")
    if use_dist:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    wall, kernel_ms, fused_ms, tm_us, c4_wall, c4_kernel_ms, wall_with_barrier = (float(x) for x in stats)
    fused_bytes = n * (2 * S * 20 + 8 + 8 + 4)            # obs + i32 actions + reward + done; no state round trip
    fused = {"what": "same steps as one launch (state resident in LDS), action tape in HBM, outputs in place",
             "steps": SEC_STEPS, "us_per_step": fused_ms * 1e3 / SEC_STEPS, "value": 2 * n * world * SEC_STEPS / (fused_ms / 1e3),
             "unit": "agent-env steps/s", "bytes_per_step": fused_bytes,
             "achieved_GBps": fused_bytes / (fused_ms / 1e3 / SEC_STEPS) / 1e9,
             "frac_of_8TBps": fused_bytes / (fused_ms / 1e3 / SEC_STEPS) / 1e9 / HBM_PEAK_GBS, **destination(2 * n * S * 20 / 1e6),
             "note": "the 52 MB in-place destination lives in the 256 MB Infinity Cache: this is on-die fabric throughput, not HBM bandwidth "
                     "(HBM's achievable rate is ~6.3 TB/s = 0.79; the HBM-proper figure of this workload is roofline.frac_hbm_proper)"}
    hbm_proper = {"what": f"the headline workload with every step's outputs written to their own time-major slot ({slots} slots, "
                          f"{slots * 2 * n * S * 20 / 1e9:.1f} GB: beyond the 256 MB Infinity Cache), one launch per step, non-temporal plane stores",
                  "steps": max(1, SEC_STEPS // slots) * slots, "us_per_launch": tm_us, "value": 2 * n * world / tm_us * 1e6, "unit": "agent-env steps/s",
                  "achieved_GBps": algorithmic_bytes(S) * n / tm_us / 1e3, "frac_of_8TBps": algorithmic_bytes(S) * n / tm_us / 1e3 / HBM_PEAK_GBS}
    c4_us = c4_kernel_ms * 1e3 / c4_steps
    config4_split = {"what": f"BASELINE config 4's split (strong scaling): asymmetric_advantages, {c4_total} kitchens in total = {c4_n} per GPU "
                             f"on {world} GPU(s), one launch per step, outputs in place",
                     "steps": c4_steps, "kitchens_per_gpu": c4_n, "us_per_launch": c4_us, "value": 2 * c4_n * world * c4_steps / c4_wall,
                     "unit": "agent-env steps/s", "scaling": "strong",
                     "achieved_GBps_per_gpu": algorithmic_bytes(c4_S) * c4_n / c4_us / 1e3,
                     "frac_of_8TBps": algorithmic_bytes(c4_S) * c4_n / c4_us / 1e3 / HBM_PEAK_GBS, **destination(2 * c4_n * c4_S * 20 / 1e6)}

    # Secondary figure, never `value`: the per-update trajectory exchange -- the one place the north star's collective meets a
    # clock.  Runs whenever a process group exists (world > 1, or OC_BENCH_FORCE_DIST=1 with one rank).
    exchange = None
    if use_dist and not args.no_secondary:
        exchange = exchange_leg(world, rank, local, backend, gen, c4_wall / c4_steps, regions)

    # Secondary figure, never `value`: policies in the loop.  8 conventions in the reference's 15 pairing
    # segments, every actor on both sides evaluated by ONE fused MFMA launch (oc_policy_forward) per step + the env launch;
    # random-weight networks of the reference shape; own engine because 15 segments do not divide 65 536.
    policy = None
    if world == 1 and not args.no_policy and not args.no_secondary and params["width"] * params["height"] <= 63:
      try:                                                 # a secondary figure must never take the headline line down with it
          from diverse_conventions_amd import FusedPairedRollout, PackedNetworks
          k = 7
          threads = n // (2 * k + 1)
          W, H = params["width"], params["height"]
          P = (W - 2) * (H - 2)
          g2 = torch.Generator().manual_seed(0)
          r = lambda *sh: torch.randn(*sh, generator=g2) * 0.1
          sds = [{"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
                  "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
                  "act.action_out.linear.weight": r(6, 64), "act.action_out.linear.bias": r(6)} for _ in range(k + 1)]
          nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=sds)
          eng2 = OvercookedEngine(params, threads * (2 * k + 1), local)
          ro = FusedPairedRollout(eng2, nets, threads, k)
          ro.step(SEC_WARMUP)
          torch.cuda.synchronize()
          psteps = SEC_STEPS
          pe0, pe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          with regions:
              pe0.record()
              for _ in range(psteps):
                  ro.plan.forward(eng2.obs, actions=ro.actions, action_logp=ro.logp, seed=0, counter=0, device_counter=ro.draw)
              pe1.record()
              torch.cuda.synchronize()
          pol_us = pe0.elapsed_time(pe1) * 1e3 / psteps
          # the whole step: a policy launch and an env launch, every launch of every step enqueued from C by one call (the env step
          # folded INTO the policy launch was built and measured in round 3 -- 116 against 60 us per step -- and lives behind the
          # diagnostics build now: DESIGN.md 9b, tools/exp_fold_ab.py)
          step_modes = {}
          ro.step(SEC_WARMUP)
          torch.cuda.synchronize()
          step_modes["two_launches"] = float("inf")
          for _ in range(2):                             # (best of two, as for the share leg below)
              with regions:
                  t1 = time.perf_counter()
                  ro.step(psteps)
                  torch.cuda.synchronize()
                  step_modes["two_launches"] = min(step_modes["two_launches"], (time.perf_counter() - t1) / psteps * 1e6)
          step_us = step_modes["two_launches"]
          # ... and what training does: the same two launches per step, every step's planes into their own time-major slot
          # (TrajectoryBuffer; the fresh planes are stored plainly while the rollout runs -- the next launch reads them)
          from diverse_conventions_amd import TrajectoryBuffer
          ctraj = TrajectoryBuffer(eng2, 200)
          ro.collect(ctraj)
          torch.cuda.synchronize()
          with regions:
              t1 = time.perf_counter()
              for _ in range(3):
                  ro.collect(ctraj)
              torch.cuda.synchronize()
              step_modes["two_launches_time_major_slots"] = (time.perf_counter() - t1) / 600 * 1e6
          del ctraj
          rows = 2 * eng2.num_kitchens
          cycles = rows / 32 * 32 * (P * (6 * nets.digits + 12) + 24 + 12)     # MFMA issue cycles (i8 conv digits + f16 two-piece products), DESIGN 9a
          cus = torch.cuda.get_device_properties(local).multi_processor_count
          bound_us = cycles / (4 * cus) / 2400.0
          policy = {"what": f"{k + 1} conventions in {2 * k + 1} pairing segments of {threads} kitchens: every actor, both sides, in one "
                            "fused MFMA launch per step + the env launch (FusedPairedRollout); random-weight reference-shape networks",
                    "steps": psteps, "kitchens": eng2.num_kitchens, "us_per_step": step_us, "value": 2 * eng2.num_kitchens / step_us * 1e6,
                    "unit": "agent-env steps/s", "us_per_step_by_mode": step_modes,
                    "checked": check_policy_in_loop(ro, eng2, params),
                    "checked_note": "after the timed steps: 24 more steps recorded into a trajectory, 4 x 64 kitchens spread over the pairing "
                                    "segments replayed through the CPU oracle with the recorded actions (obs, reward, done bit-identical)",
                    "policy_kernel": {"kernel": f"oc_policy_kernel<{nets.digits} digits>", "us_per_launch": pol_us, "network_evaluations": rows,
                                      "useful_TFLOPs": rows * 2 * (P * 180 * 32 + P * 32 * 64 + 64 * 64 + 64 * 6) / pol_us * 1e-6,
                                      "bound": "mfma", "issue_bound_us_at_2.4GHz": bound_us, "frac": bound_us / pol_us,
                                      "note": "matrix cycles per 32 rows: 6912 (three base-256 i8 conv digits + f16 two-piece products -- hi hi, hi lo, lo hi -- for "
                                              "the dense layers; 10 368 with three bf16 pieces in round 3, 11 520 in round 2, 23 040 with f32 MFMAs in round 1); the VALU "
                                              "work between the layers does not overlap the matrix pipe on a gfx950 SIMD, DESIGN.md 9a"}}
          eng2.close()
          # the same rollout with the kitchens per pairing segment a multiple of 32 (aligned_threads: what INTEGRATION.md recommends --
          # the policy kernel evaluates 32 rows per block): 15 x 4352 kitchens, no ragged block per segment
          from diverse_conventions_amd import aligned_threads
          th_a = aligned_threads(n, k)
          eng2a = OvercookedEngine(params, th_a * (2 * k + 1), local)
          ro_a = FusedPairedRollout(eng2a, nets, th_a, k)
          ro_a.step(SEC_WARMUP)
          torch.cuda.synchronize()
          with regions:
              pe0.record()
              for _ in range(psteps):
                  ro_a.plan.forward(eng2a.obs, actions=ro_a.actions, action_logp=ro_a.logp, seed=0, counter=0, device_counter=ro_a.draw)
              pe1.record()
              torch.cuda.synchronize()
          pol_a_us = pe0.elapsed_time(pe1) * 1e3 / psteps
          step_a_us = float("inf")
          for _ in range(2):
              with regions:
                  t1 = time.perf_counter()
                  ro_a.step(psteps)
                  torch.cuda.synchronize()
                  step_a_us = min(step_a_us, (time.perf_counter() - t1) / psteps * 1e6)
          rows_a = 2 * eng2a.num_kitchens
          policy["aligned_segments"] = {
              "what": f"the same rollout at aligned_threads({n}, {k}) = {th_a} kitchens per pairing segment ({eng2a.num_kitchens} kitchens): segments of whole "
                      "32-row blocks",
              "kitchens": eng2a.num_kitchens, "us_per_step": step_a_us, "value": 2 * eng2a.num_kitchens / step_a_us * 1e6, "unit": "agent-env steps/s",
              "policy_kernel_us_per_launch": pol_a_us, "network_evaluations": rows_a,
              "policy_kernel_frac_of_issue_bound": bound_us * rows_a / rows / pol_a_us, "checked": check_policy_in_loop(ro_a, eng2a, params)}
          eng2a.close()
      except Exception as exc:                             # noqa: BLE001
        policy = {"error": f"{type(exc).__name__}: {exc}"}

    # Secondary figure, never `value`: BASELINE config 4 with the policies in the loop, one GPU's share on the 8-GPU node the config
    # names -- asymmetric_advantages, 8184 kitchens, 16 conventions in 31 pairing segments, actors and one critic per segment on both
    # sides in ONE policy launch per step + the env launch (what DESIGN.md 7's strong-scaling prediction is built from)
    share = None
    if policy is not None and "error" not in policy:
      try:
          from diverse_conventions_amd import FusedPairedRollout, PackedNetworks
          sk, sthreads = 15, (65536 // 8) // 31
          sG = 2 * sk + 1
          sp = get_base_layout_params("asymmetric_advantages", 200)
          sW, sH = sp["width"], sp["height"]
          sP = (sW - 2) * (sH - 2)
          g3 = torch.Generator().manual_seed(1)
          r3 = lambda *sh: torch.randn(*sh, generator=g3) * 0.1
          base3 = lambda: {"base.cnn.cnn.0.weight": r3(32, 20, 3, 3), "base.cnn.cnn.0.bias": r3(32), "base.cnn.cnn.3.weight": r3(64, 32 * sP),
                           "base.cnn.cnn.3.bias": r3(64), "base.cnn.cnn.5.weight": r3(64, 64), "base.cnn.cnn.5.bias": r3(64)}
          actors3 = [dict(base3(), **{"act.action_out.linear.weight": r3(6, 64), "act.action_out.linear.bias": r3(6)}) for _ in range(sk + 1)]
          critics3 = [dict(base3(), **{"v_out.weight": r3(1, 64), "v_out.bias": r3(1)}) for _ in range(sG)]
          nets3 = PackedNetworks.from_state_dicts((sW, sH, 20), "cuda", actors=actors3, critics=critics3)
          eng3 = OvercookedEngine(sp, sthreads * sG, local)
          ro3 = FusedPairedRollout(eng3, nets3, sthreads, sk, critic_nets=[[sk + 1 + g for g in range(sG)]] * 2)
          ro3.step(SEC_WARMUP)
          torch.cuda.synchronize()
          share_us = float("inf")
          for _ in range(2):                             # (best of two: 2 x SEC_STEPS launches of ~20 us each are enqueued by the host as they run,
              with regions:                              #  and a descheduled host thread shows up as GPU idle time -- one driver-style run read 79.8 us)
                  t3 = time.perf_counter()
                  ro3.step(SEC_STEPS)
                  torch.cuda.synchronize()
                  share_us = min(share_us, (time.perf_counter() - t3) / SEC_STEPS * 1e6)
          share = {"what": f"BASELINE config 4, one GPU's share of 8: asymmetric_advantages, {eng3.num_kitchens} kitchens, {sk + 1} conventions in {sG} "
                           f"pairing segments of {sthreads} kitchens, every actor and one critic per segment on both sides in ONE fused MFMA launch "
                           "per step + the env launch, all steps enqueued by one call; random-weight reference-shape networks",
                   "steps": SEC_STEPS, "kitchens": eng3.num_kitchens, "network_evaluations_per_step": 4 * eng3.num_kitchens, "us_per_step": share_us,
                   "value": 2 * eng3.num_kitchens / share_us * 1e6, "unit": "agent-env steps/s",
                   "checked": check_policy_in_loop(ro3, eng3, sp),
                   "note": "stretches of 4 k + 1 blocks on the four-wavefront workgroups share their last block (cooperative blocks, DESIGN.md 9a): "
                           "62.5 us per step without"}
          eng3.close()
      except Exception as exc:                             # noqa: BLE001
        share = {"error": f"{type(exc).__name__}: {exc}"}

    # Secondary figure, never `value`: BASELINE config 2 -- cramped_room, 4096 kitchens, step + encode kernel only: per launch
    # (launch-bound: 4 MB per step) and as the K-steps-per-launch rollout
    config2 = config4_share_env_only = None
    if world == 1 and not args.no_secondary:
        c2_n = 4096
        c2_params = get_base_layout_params("cramped_room", 200)
        eng_c2 = OvercookedEngine(c2_params, c2_n, local)
        act_c2 = torch.randint(0, 6, (t_act, 2, c2_n), generator=gen, device="cuda", dtype=torch.int32)
        eng_c2.rollout(act_c2, SEC_WARMUP)
        with regions:
            c2_us = eng_c2.rollout_timed(act_c2, SEC_STEPS) * 1e3 / SEC_STEPS
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng_c2.rollout_fused(SEC_WARMUP, actions=act_c2)
        with regions:
            ev0.record()
            eng_c2.rollout_fused(SEC_STEPS, actions=act_c2)
            ev1.record()
            torch.cuda.synchronize()
        c2_fused_us = ev0.elapsed_time(ev1) * 1e3 / SEC_STEPS
        c2_bytes = algorithmic_bytes(20) * c2_n
        config2 = {"what": "BASELINE config 2: cramped_room, 4096 parallel kitchens, random-action rollout, step + encode kernel only",
                   "steps": SEC_STEPS, "kitchens": c2_n, "us_per_launch": c2_us, "value": 2 * c2_n / c2_us * 1e6, "unit": "agent-env steps/s",
                   "bytes_per_launch": c2_bytes, "achieved_GBps": c2_bytes / c2_us / 1e3, "frac_of_8TBps": c2_bytes / c2_us / 1e3 / HBM_PEAK_GBS,
                   **destination(2 * c2_n * 20 * 20 / 1e6), "kernel_build": eng_c2.kernel_build_info,
                   "fused_rollout_us_per_step": c2_fused_us, "fused_rollout_value": 2 * c2_n / c2_fused_us * 1e6,
                   "note": "one launch per step is launch-bound at this size (a wavefront's own critical path is ~3.4 us, DESIGN.md 5); the "
                           "K-steps-per-launch rollout has no launch boundary per step"}
        eng_c2.close()
        # ... and one GPU's share of BASELINE config 4 as an env step only: asymmetric_advantages x 8192 kitchens (what each of 8 GPUs
        # steps per launch; VERDICT r05 item 6 asked for it in the line)
        c4s_params = get_base_layout_params("asymmetric_advantages", 200)
        eng_c4s = OvercookedEngine(c4s_params, 8192, local)
        act_c4s = torch.randint(0, 6, (t_act, 2, 8192), generator=gen, device="cuda", dtype=torch.int32)
        eng_c4s.rollout(act_c4s, SEC_WARMUP)
        with regions:
            c4s_us = eng_c4s.rollout_timed(act_c4s, SEC_STEPS) * 1e3 / SEC_STEPS
        c4s_bytes = algorithmic_bytes(45) * 8192
        config4_share_env_only = {"what": "BASELINE config 4, one GPU's share of 8, env step only: asymmetric_advantages x 8192 kitchens, one launch per step, in place",
                                  "steps": SEC_STEPS, "kitchens": 8192, "us_per_launch": c4s_us, "value": 2 * 8192 / c4s_us * 1e6, "unit": "agent-env steps/s",
                                  "bytes_per_launch": c4s_bytes, "achieved_GBps": c4s_bytes / c4s_us / 1e3, "frac_of_8TBps": c4s_bytes / c4s_us / 1e3 / HBM_PEAK_GBS,
                                  **destination(2 * 8192 * 45 * 20 / 1e6), "kernel_build": eng_c4s.kernel_build_info,
                                  "note": "launch-bound: ~2.5 us of dependent-launch cost + a ~3 us per-wavefront chain; round 6: 4 kitchens per group in "
                                          "workgroups of eight wavefronts (6.5-7.1 us with 8 per group in one-wavefront workgroups)"}
        eng_c4s.close()

    # Secondary figure, never `value`: kitchens with other player counts (the reference's `num_players`: multiplayer_schelling with 4 and 3
    # players, simple_single), one launch per step on the N-player group kernel (csrc/oc_np_kernels.h), outputs in place
    nplayer = None
    if world == 1 and not args.no_secondary:
      try:
          nplayer = {"what": "kitchens with 4 / 3 / 1 players (multiplayer_schelling, simple_single), 65536 kitchens, one launch per step "
                             "(transition + encoding of every viewer), outputs in place; bytes per kitchen-step = P*S*(5P+10) obs + 2 records + 4P actions + 4P reward + 4 done",
                     "steps": SEC_STEPS // 2, "cases": []}
          for np_layout, np_max in (("multiplayer_schelling", None), ("multiplayer_schelling", 3), ("simple_single", None)):
              npp = get_base_layout_params(np_layout, 200, max_num_players=np_max)
              nP, nS, nn = npp["num_players"], npp["width"] * npp["height"], 65536
              eng_np = OvercookedEngine(npp, nn, local)
              act_np = torch.randint(0, 6, (t_act, nP, nn), generator=gen, device="cuda", dtype=torch.int32)
              eng_np.rollout(act_np, 50)
              with regions:
                  np_us = min(eng_np.rollout_timed(act_np, SEC_STEPS // 2) for _ in range(2)) * 1e3 / (SEC_STEPS // 2)
              np_bytes = nn * (nP * nS * (5 * nP + 10) + 2 * eng_np.record_bytes + 8 * nP + 4)
              nplayer["cases"].append({"layout": np_layout, "players": nP, "kitchens": nn, "us_per_launch": np_us, "value": nP * nn / np_us * 1e6,
                                       "unit": "agent-env steps/s", "bytes_per_launch": np_bytes, "achieved_GBps": np_bytes / np_us / 1e3,
                                       "frac_of_8TBps": np_bytes / np_us / 1e3 / HBM_PEAK_GBS, **destination(nn * nP * nS * (5 * nP + 10) / 1e6),
                                       "kernel_build": eng_np.kernel_build_info})
              eng_np.close()
              del act_np
      except Exception as exc:                             # noqa: BLE001
        nplayer = {"error": f"{type(exc).__name__}: {exc}"}

    # Secondary figure, never `value`: the mixed-play collection: 329 x 199 = 65 471 kitchens, two episodes of
    # L = 200 steps, per step ONE gathered policy launch (every row evaluated once, by the network the coin gave it) + the env
    # launch, all 2L steps enqueued by one call; checked: sampled kitchens replayed through the CPU oracle with the recorded actions
    mixed = None
    if policy is not None and "error" not in policy:
      try:
          import numpy as np
          from diverse_conventions_amd import MixedPlayRollout, PackedNetworks
          from oracle import oracle
          mL, mreps = 200, 329
          mn = mreps * (mL - 1)
          mparams = get_base_layout_params("cramped_room", mL)
          mW, mH = mparams["width"], mparams["height"]
          mP = (mW - 2) * (mH - 2)
          g4 = torch.Generator().manual_seed(2)
          r4 = lambda *sh: torch.randn(*sh, generator=g4) * 0.1
          mbase = lambda: {"base.cnn.cnn.0.weight": r4(32, 20, 3, 3), "base.cnn.cnn.0.bias": r4(32), "base.cnn.cnn.3.weight": r4(64, 32 * mP),
                           "base.cnn.cnn.3.bias": r4(64), "base.cnn.cnn.5.weight": r4(64, 64), "base.cnn.cnn.5.bias": r4(64)}
          mactors = [dict(mbase(), **{"act.action_out.linear.weight": r4(6, 64), "act.action_out.linear.bias": r4(6)}) for _ in range(2)]
          mcritics = [dict(mbase(), **{"v_out.weight": r4(1, 64), "v_out.bias": r4(1)}) for _ in range(2)]
          mnets = PackedNetworks.from_state_dicts((mW, mH, 20), "cuda", actors=mactors, critics=mcritics)
          meng = OvercookedEngine(mparams, mn, local)
          mro = MixedPlayRollout(meng, mnets, mL, main=(0, 2), partner=(1, 3), seed=3)
          mro.collect()
          torch.cuda.synchronize()
          with regions:
              t5 = time.perf_counter()
              mro.collect()
              torch.cuda.synchronize()
              m_s = time.perf_counter() - t5
          # outside the timed region: 3 x 48 kitchens (first, middle, last roles of the staircase) through the oracle
          m_ok, per = True, 48
          for f in (0, (mn - per) // 2, mn - per):
              ora = oracle.OracleVecEnv(mparams, per)
              sl = slice(f, f + per)
              m_ok = m_ok and np.array_equal(mro.traj.obs[0][:, sl].cpu().numpy(), ora.encode())
              a_all = mro.traj.actions[:, :, sl].cpu().numpy()
              for t in range(2 * mL):
                  oobs, orew, odone = ora.step(np.ascontiguousarray(a_all[t]))
                  if t % 25 == 0 or t in (mL - 1, mL, 2 * mL - 1):
                      m_ok = m_ok and np.array_equal(mro.traj.obs[t + 1][:, sl].cpu().numpy(), oobs)
                  m_ok = m_ok and np.array_equal(mro.traj.reward[t][:, sl].cpu().numpy(), orew) and np.array_equal(mro.traj.done[t][sl].cpu().numpy(), odone)
          mixed = {"what": f"mixed-play collection: cramped_room, {mn} kitchens = {mreps} x {mL - 1} roles of the staircase, two episodes of L = {mL} "
                           "steps, per step one gathered policy launch (2 actors + 2 critics, every row evaluated once) + the env launch, one call",
                   "steps": 2 * mL, "kitchens": mn, "collect_ms": m_s * 1e3, "us_per_step": m_s / (2 * mL) * 1e6,
                   "value": 2 * mn * 2 * mL / m_s, "unit": "agent-env steps/s", "checked": bool(m_ok),
                   "checked_note": "3 x 48 kitchens (first / middle / last staircase roles) replayed through the CPU oracle with the recorded "
                                   "actions: rewards and dones at every step, observations at 20 steps incl. both episode ends"}
          meng.close()
          del mro
      except Exception as exc:                             # noqa: BLE001
        mixed = {"error": f"{type(exc).__name__}: {exc}"}

    # At N = 1 the exchange leg still runs, as a one-rank RCCL call on device buffers (a one-rank all-gather is a copy through RCCL's
    # channels: it says nothing about xGMI, but the collective, the wire buffer and the receive side are the ones an 8-GPU run uses)
    if exchange is None and world == 1 and not args.no_secondary and not use_dist and torch.cuda.is_available():
      try:
          import socket
          with socket.socket() as sock:
              sock.bind(("127.0.0.1", 0))
              port = sock.getsockname()[1]
          dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", local))
          try:
              exchange = exchange_leg(1, 0, local, "nccl", gen, c4_wall / c4_steps, regions)
          finally:
              dist.destroy_process_group()
      except Exception as exc:                             # noqa: BLE001
        exchange = {"error": f"{type(exc).__name__}: {exc}"}

    drop_in = eval_1000 = None
    if world == 1 and not args.no_secondary and not args.no_policy:
        try:
            drop_in, eval_1000 = drop_in_and_eval_legs(local, regions, gen)
        except Exception as exc:                           # noqa: BLE001
            drop_in = {"error": f"{type(exc).__name__}: {exc}"}

    build_info = eng.kernel_build_info
    if rank == 0:
        bytes_per_launch = algorithmic_bytes(S) * n
        per_launch_s = kernel_ms / 1e3 / args.steps
        achieved = bytes_per_launch / per_launch_s / 1e9
        # HBM traffic per launch comes from PMC passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs of this very
        # command, see profiles/README.md), not from this process: it is read from the committed summary and labelled so
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            entry = tj.get(f"{args.layout}:{n}", {})
            traffic = entry.get("bytes_per_launch")
            if traffic is not None:
                traffic_source = f"profiles/hbm_traffic.json ({entry.get('measured', 'round unknown')}): {entry.get('source', '')}"
        # the rocprofv3 --kernel-trace --stats average of the headline kernel on this very workload (committed summary of a
        # 2000-step run, profiles/kernel_us.json names the csv): the short driver run and the long profile in one record
        kernel_us_profile, kernel_us_profile_source = None, None
        kpath = os.path.join(ROOT, "profiles", "kernel_us.json")
        if os.path.exists(kpath):
            with open(kpath) as f:
                kj = json.load(f).get(f"{args.layout}:{n}", {})
            kernel_us_profile, kernel_us_profile_source = kj.get("avg_us"), kj.get("source")
        out = {
            "metric": "agent-env steps/sec at 64k parallel kitchens, 1/2/4/8 MI355X",
            "value": 2 * n * world * args.steps / wall,
            "unit": "agent-env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "launcher": launcher,
            "timing": "per rank: barrier + synchronize, clock, K steps, synchronize, clock, barrier; ms_per_step and value from the MAX over ranks",

            "ms_per_step": wall * 1e3 / args.steps,
            # what the wall clock adds to the kernels' own time over the K-step region (enqueue, the two syncs, the clock reads): a 20-step
            # run is 220 us of kernels, so tens of microseconds of host cost move `value` by tens of percent from box to box (profiles/README.md)
            "kernel_ms_total": kernel_ms, "host_fixed_us": wall * 1e6 - kernel_ms * 1e3,
            "value_kernel_time": 2 * n * world * args.steps / (kernel_ms / 1e3),
            "wall_with_barrier_ms": wall_with_barrier * 1e3,
            "wall_with_barrier_note": "rounds 1-4 timed barrier-to-barrier (the closing all-reduce inside the clock); `value` stops each rank's clock "
                                      "at its own synchronize -- compare like with like across rounds",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.layout}, {n} kitchens per GPU, 2 agents, uniform random joint actions "
                                   f"(i32, resident in HBM), horizon 200 with in-step auto-reset, fused step+encode, "
                                   f"one launch per step",
                       "layout": args.layout, "kitchens_per_gpu": n, "parallelism": f"kitchen-shard x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "frac_hbm_proper": None if args.no_secondary else hbm_proper["frac_of_8TBps"],
                         **destination(2 * n * S * 20 / 1e6),
                         "bound_note": "the reference's data-plane contract is an in-place output buffer (52 MB at this size): it is overwritten in the "
                                       "256 MB Infinity Cache, so `frac` is the step's throughput against the HBM peak, not measured HBM bandwidth; with "
                                       "every step in its own time-major slot (3.4 GB, nothing absorbed on-die) the same launch reaches `frac_hbm_proper`; "
                                       "HBM's achievable rate is ~0.79 of the peak",
                         "kernel": "oc_jit_step_i32_v16" if "specialised" in build_info else "oc_step_kernel<STEP,i32,vec16>",
                         "kernel_build": build_info, "bytes_per_launch": bytes_per_launch,
                         "us_per_launch": per_launch_s * 1e6, "kernel_us_profile": kernel_us_profile,
                         "kernel_us_profile_source": kernel_us_profile_source},
        }
        if not args.no_secondary:
            # the same roofline object for the HBM-proper leg: nothing of its 3.4 GB of destinations fits the 256 MB Infinity Cache
            out["roofline_hbm_proper"] = {"bound": "hbm", "achieved": hbm_proper["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": hbm_proper["frac_of_8TBps"], "us_per_launch": tm_us, "bytes_per_launch": bytes_per_launch,
                                          "kernel": "oc_jit_step_i32_v16_nt (time-major slots, non-temporal plane stores)", "traffic": None}
            out["hbm_proper"] = hbm_proper
            out["config4_split"] = config4_split
            if exchange is not None:
                out["exchange"] = exchange
            if c3 is not None:
                out["config3_one_launch"] = c3
            out["fused_rollout"] = fused
        if config2 is not None:
            out["config2"] = config2
            out["config4_share_env_only"] = config4_share_env_only
        if nplayer is not None:
            out["nplayer"] = nplayer
        if mixed is not None:
            out["mixed_play"] = mixed
        if policy is not None:
            out["policy_in_loop"] = policy
        if share is not None:
            out["config4_share_policy_in_loop"] = share
        if drop_in is not None:
            out["drop_in_step"] = drop_in
        if eval_1000 is not None:
            out["eval_1000"] = eval_1000
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, n)
        out["cold_start"] = {"headline_oc_create_ms": headline_create_ms, "oc_create_ms_8192_kitchens": create_ms,
                             "note": "wall time of OvercookedEngine(...) = oc_create: module load (ahead-of-time code object, else disk cache, else a "
                                     "hiprtc build of about a second), device buffers, tables, start-state fill and one device synchronize"}
        out["jit_builds_inside_timed_regions"] = int(regions.inside)
        out["jit_builds"] = {"before_the_headline_region": int(builds_before_timing), "whole_process": regions.total_builds(),
                             "timed_regions": regions.regions,
                             "note": "hiprtc compilations (about a second each): 0 in the whole process when the ahead-of-time code objects "
                                     "(csrc/aot, built by __graft_entry__.build()) match the box; never inside a timed region"}
        if regions.inside:
            print(f"bench.py: {regions.inside} hiprtc build(s) ran INSIDE a timed region -- a figure of this line is a compile time", file=sys.stderr)
        print(json.dumps(out), file=real_stdout, flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
