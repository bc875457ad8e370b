"""Fused MFMA policy inference (oc_policy_forward) against the reference's R_Actor / R_Critic.

Oracle chain: tests/golden/policy_cases*.npz hold outputs of the reference modules themselves (two grid shapes);
oracle/policy_oracle.reference_forward is the numpy restatement pinned to them; emulate_packed replays the kernel's
lane-level dataflow on the packed weight image (CPU).  Floating point: the kernel's head outputs must be within
TOL = 2e-6 absolute of the reference's float32 forward with 3 (the default: 24 bits) or 4 base-256 weight digits (the
summation order differs from torch's GEMM; the dense layers are exact bf16 piece products accumulated in f32, the conv is exact
integer arithmetic on the digits); deterministic actions equal."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from diverse_conventions_amd import native
from diverse_conventions_amd.fused_policy import (ACTOR_HEAD, CRITIC_HEAD, TILE_ACT, PackedNetworks, PolicyPlan, draw_uniform,
                                                  pack_host, packed_bytes, plan_tiles)
from oracle.policy_oracle import emulate_packed, log_softmax, parse_image, reference_forward

TOL = 2e-6
TOL_LOGP = 4e-6          # log-softmax adds one expf/logf round trip


def cases(name="policy_cases.npz"):
    z = np.load(os.path.join(GOLDEN, name))
    get = lambda pre: {k[len(pre):]: z[k] for k in z.files if k.startswith(pre) and "weight" in k or k.startswith(pre) and "bias" in k}
    return z, get


def random_net(W, H, n_out, head, seed):
    rng = np.random.default_rng(seed)
    P = (W - 2) * (H - 2)
    r = lambda *s: (rng.standard_normal(s) * 0.12).astype(np.float32)
    return {"base.cnn.cnn.0.weight": r(32, 20, 3, 3), "base.cnn.cnn.0.bias": r(32), "base.cnn.cnn.3.weight": r(64, 32 * P),
            "base.cnn.cnn.3.bias": r(64), "base.cnn.cnn.5.weight": r(64, 64), "base.cnn.cnn.5.bias": r(64),
            head + ".weight": r(n_out, 64), head + ".bias": r(n_out)}


# ------------------------------------------------------------------------------------------- CPU: oracle, packing
def test_oracle_matches_reference_outputs():
    z, get = cases()
    for i in range(3):
        out = reference_forward(get(f"a{i}."), z["obs"], ACTOR_HEAD)
        assert np.abs(log_softmax(out) - z[f"a{i}.logits"]).max() < 1e-6
        assert np.array_equal(out.argmax(-1)[:, None], z[f"a{i}.det_action"])
    for i in range(2):
        assert np.abs(reference_forward(get(f"c{i}."), z["obs"], CRITIC_HEAD) - z[f"c{i}.values"]).max() < 1e-6
    z, get = cases("policy_cases_unident_s.npz")
    assert np.abs(log_softmax(reference_forward(get("a0."), z["obs"], ACTOR_HEAD)) - z["a0.logits"]).max() < 1e-6
    assert np.abs(reference_forward(get("c0."), z["obs"], CRITIC_HEAD) - z["c0.values"]).max() < 1e-6


@pytest.mark.parametrize("digits,tol", [(4, 2e-6), (3, 2e-6), (2, 6e-5)])
def test_packed_image_emulation_matches_reference(digits, tol):
    """Host packer + lane-level emulation of the kernel's MFMA dataflow == reference outputs (no GPU involved)."""
    z, get = cases()
    blob = pack_host(get("a1."), (5, 4, 20), ACTOR_HEAD, digits)
    meta, _ = parse_image(blob)
    assert meta == dict(W=5, H=4, n_out=6, digits=digits, P=6) and len(blob) == packed_bytes(5, 4, digits)
    out = emulate_packed(blob, z["obs"][:48].reshape(48, -1))
    assert np.abs(log_softmax(out) - z["a1.logits"][:48]).max() < tol
    if digits >= 3:
        blob = pack_host(get("c1."), (5, 4, 20), CRITIC_HEAD, digits)
        assert np.abs(emulate_packed(blob, z["obs"][100:133].reshape(33, -1)) - z["c1.values"][100:133]).max() < tol


def test_packed_image_emulation_second_grid():
    z, get = cases("policy_cases_unident_s.npz")
    for digits in (3, 4):
        out = emulate_packed(pack_host(get("a0."), (9, 5, 20), ACTOR_HEAD, digits), z["obs"][:33].reshape(33, -1))
        assert np.abs(log_softmax(out) - z["a0.logits"][:33]).max() < 2e-6


def test_packed_image_emulation_large_conv_biases():
    """Biases far larger than the channel's weights coarsen that channel's fixed-point unit (they must fit 2^30 units of the
    32-bit accumulator); the result stays f32-class because the channel's value is then dominated by the bias."""
    rng = np.random.default_rng(9)
    sd = random_net(5, 4, 6, ACTOR_HEAD, 11)
    amax = np.abs(sd["base.cnn.cnn.0.weight"]).reshape(32, -1).max(1)
    sd["base.cnn.cnn.0.bias"][:8] = amax[:8] * np.array([5, -7, 40, -60, 300, -900, 1e4, 3], dtype=np.float32)
    obs = ((rng.random((40, 5, 4, 20)) < 0.15) * rng.integers(1, 21, (40, 5, 4, 20))).astype(np.int8)
    want = reference_forward(sd, obs, ACTOR_HEAD)
    for digits in (3, 4):
        got = emulate_packed(pack_host(sd, (5, 4, 20), ACTOR_HEAD, digits), obs.reshape(40, -1))
        assert np.abs(got - want).max() < 4e-6 * max(1.0, np.abs(want).max())


def test_conv_digits_reconstruct_weights():
    sd = random_net(5, 4, 6, ACTOR_HEAD, 3)
    sd["base.cnn.cnn.0.weight"][7] = 0                      # an all-zero channel must not divide by zero
    sd["base.cnn.cnn.0.bias"][31] = 50 * np.abs(sd["base.cnn.cnn.0.weight"][31]).max()     # a bias that outgrows the 32-bit accumulator units
    for digits in (1, 2, 3, 4):
        _, sec = parse_image(pack_host(sd, (5, 4, 20), ACTOR_HEAD, digits))
        conv, scales, bias_units = sec["conv"].astype(np.int64), sec["conv_scale"], sec["conv_bias"].astype(np.int64)
        assert np.abs(conv[:, 0]).max() <= 127 and (digits == 1 or (conv[:, 1:].min() >= -128 and conv[:, 1:].max() <= 127))
        q = sum(conv[:, d] * 256 ** (digits - 1 - d) for d in range(digits))          # [6][64][16]
        w, b = sd["base.cnn.cnn.0.weight"], sd["base.cnn.cnn.0.bias"]
        for o in (0, 7, 31):
            lane_h0, lane_h1 = o, 32 + o
            reg, h = [(r, hh) for r in range(16) for hh in range(2) if (r & 3) + 8 * (r >> 2) + 4 * hh == o][0]
            scale = np.float64(scales[h][reg])
            # one unit of the lowest digit: absmax / qmax, unless the bias needs coarser units to fit 2^30 of them
            want_scale = max(np.abs(w[o]).max() / (126 * 256 ** (digits - 1)), abs(b[o]) / 2 ** 30) or 1.0
            assert abs(scale - want_scale) <= 1e-7 * want_scale
            assert abs(bias_units[h][reg]) <= 2 ** 30 and abs(bias_units[h][reg] * scale - b[o]) <= 0.51 * scale + 1e-7 * abs(b[o])
            for kx in range(3):
                k = np.concatenate([q[kx * 2 + s, lane] for s in range(2) for lane in (lane_h0, lane_h1)])   # k 0..63
                want = w[o, :, kx, :].T.reshape(-1)                                     # (ky, c) order, 60 values
                assert np.all(k[60:] == 0)
                assert np.abs(k[:60] * scale - want).max() <= 0.51 * scale + 2e-7 * np.abs(w[o]).max() + 1e-12


def test_abi_argument_checks_and_tile_plan():
    L = native.lib()
    S = native.OcPolicyShape
    assert L.oc_policy_packed_bytes(ctypes.byref(S(5, 4, 64, 6, 4))) == packed_bytes(5, 4, 4) > 0
    assert L.oc_policy_packed_bytes(ctypes.byref(S(5, 4, 128, 6, 4))) < 0 and b"hidden_size" in L.oc_last_error()
    assert L.oc_policy_packed_bytes(ctypes.byref(S(2, 4, 64, 6, 4))) < 0
    assert L.oc_policy_packed_bytes(ctypes.byref(S(5, 4, 64, 9, 4))) < 0
    assert L.oc_policy_packed_bytes(ctypes.byref(S(5, 4, 64, 6, 5))) < 0
    assert L.oc_policy_forward(ctypes.byref(S(5, 4, 64, 6, 4)), None, 1, None, 10, None, 1, None, None, None, None, 0, 0, 0, None, None, None) == -1
    t = plan_tiles([(0, 300, 2, TILE_ACT), (300, 64, 5, 0), (364, 0, 1, 0)], 128).numpy()
    assert t.tolist() == [[0, 128, 2, 1], [128, 128, 2, 1], [256, 44, 2, 1], [300, 64, 5, 0]]
    with pytest.raises(native.OvercookedNativeError):
        plan_tiles([(0, 10, 0, 0)], 100)                     # not a multiple of 32
    # the multi-grid entry refuses malformed calls before touching the GPU
    M = native.OcPolicyMultiSlot
    buf = (ctypes.c_int32 * 64)()
    tiles = ctypes.cast((ctypes.addressof(buf) + 15) // 16 * 16, ctypes.c_void_p)
    slots = (M * 2)()
    assert L.oc_policy_forward_multi(None, 1, tiles, 4, 1, 4, 0, 0, None, None) == -1
    assert L.oc_policy_forward_multi(slots, 0, tiles, 4, 1, 4, 0, 0, None, None) == -1 and b"slots" in L.oc_last_error()
    assert L.oc_policy_forward_multi(slots, 9, tiles, 4, 1, 4, 0, 0, None, None) == -1
    assert L.oc_policy_forward_multi(slots, 1, tiles, 4, 3, 4, 0, 0, None, None) == -1 and b"multiple" in L.oc_last_error()
    assert L.oc_policy_forward_multi(slots, 1, tiles, 4, 1, 4, 0, 0, None, None) == -1          # slot 0 has no shape


def _plan_multi(segs, costs, workgroups):
    import ctypes
    seg = np.asarray(segs, dtype=np.int64).reshape(-1, 5)
    cols = [np.ascontiguousarray(seg[:, 0]), np.ascontiguousarray(seg[:, 1])] + [np.ascontiguousarray(seg[:, j].astype(np.int32)) for j in (2, 3, 4)]
    cost = np.asarray(costs, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    per, waves = ctypes.c_int32(0), ctypes.c_int32(0)
    L = native.lib()
    n = L.oc_policy_plan_tiles_multi(*[p(c) for c in cols], p(cost), len(costs), len(seg), workgroups, None, 0, ctypes.byref(per), ctypes.byref(waves))
    if n < 0:
        return int(n), None, 0
    tiles = np.zeros((int(n), 4), dtype=np.int32)
    assert L.oc_policy_plan_tiles_multi(*[p(c) for c in cols], p(cost), len(costs), len(seg), workgroups, p(tiles), int(n), ctypes.byref(per),
                                        ctypes.byref(waves)) == n
    wgs = workgroups * (4 if waves.value == 1 else 1)
    return int(n), tiles.reshape(wgs if per.value > 1 or n == wgs else -1, -1, 4), per.value


def test_multi_grid_tile_plan_balances_cost_and_covers_every_row():
    """oc_policy_plan_tiles_multi (host only): with many short segments (config 5's shape: 63 segments of 130 rows per side, an actor and
    a critic each, five grids of different block cost) every workgroup gets an equal share of the cost and every row of every
    segment is covered exactly once; with two workgroups per segment or more the plan is one tile per workgroup."""
    costs = [8.5, 23.5, 11.5, 11.5, 20.5]
    segs = []
    for slot in range(5):
        for side in range(2):
            for g in range(63):
                segs.append((side * 8190 + g * 130, 130, g % 33, TILE_ACT, slot))
                segs.append((side * 8190 + g * 130, 130, 33 + g, 0, slot))
    n, tiles, per = _plan_multi(segs, costs, 256)                             # 5-block segments: planned as wave tiles, 4 x 256 stretches
    assert per >= 1 and n == 1024 * per and tiles.shape == (1024, per, 4)
    live = tiles[:, :, 1] > 0
    load = ((tiles[:, :, 1] + 31) // 32 * np.asarray(costs)[tiles[:, :, 3] >> 16] * live).sum(1)
    assert load.max() - load.min() <= 2 * max(costs) and load.min() > load.mean() - max(costs)
    covered = {}
    for r0, cnt, net, fl in tiles[live]:
        key = (int(fl) >> 16, int(net), int(fl) & 0xFFFF)
        covered.setdefault(key, []).append((int(r0), int(cnt)))
    for r0, cnt, net, fl, slot in segs:
        spans = sorted(sp for sp in covered[(slot, net, fl)] if r0 <= sp[0] < r0 + cnt)
        assert spans[0][0] == r0 and spans[-1][0] + spans[-1][1] == r0 + cnt and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(sp[0] % 32 == r0 % 32 for sp in spans)                     # cuts fall on the segment's own 32-row blocks
    # few long segments: one tile per workgroup, workgroups in proportion to cost
    n2, tiles2, per2 = _plan_multi([(0, 8192, 0, TILE_ACT, 0), (0, 8192, 0, TILE_ACT, 1)], costs[:2], 256)
    assert per2 == 1 and n2 <= 256
    t2 = tiles2.reshape(-1, 4)
    share = np.bincount(t2[:, 3] >> 16, minlength=2)
    assert abs(share[1] / share[0] - 23.5 / 8.5) < 0.2
    # argument errors are reported, not crashed on
    assert _plan_multi([(0, 10, 0, 0, 5)], costs, 256)[0] < 0 and "slot" in native.lib().oc_last_error().decode()
    assert _plan_multi([(0, 10, 0, 0, 0)], [0.0], 256)[0] < 0


def test_merge_segments_joins_touching_ranges_of_one_network():
    from diverse_conventions_amd.fused_policy import merge_segments
    got = merge_segments([(0, 10, 0, 1), (0, 10, 5, 0), (10, 10, 0, 1), (10, 10, 6, 0), (20, 5, 1, 1), (40, 10, 0, 1), (30, 10, 0, 1)])
    assert got == [(0, 20, 0, 1), (30, 20, 0, 1), (0, 10, 5, 0), (10, 10, 6, 0), (20, 5, 1, 1)]


def test_recurrent_state_dicts_are_refused_before_any_gpu_work():
    import inspect
    from diverse_conventions_amd import fused_policy
    src = inspect.getsource(fused_policy.PackedNetworks.pack)
    assert src.index('startswith("rnn.")') < src.index("oc_policy_pack")        # refused before the device packer runs
    nets = object.__new__(PackedNetworks)                                        # no GPU needed for the refusal
    with pytest.raises(NotImplementedError):
        nets.pack(0, {"rnn.rnn.weight_ih_l0": np.zeros((192, 64), np.float32)})


def test_uniform_hash_is_open_interval_and_spread():
    u = np.array([draw_uniform(7, c, r, a) for c in range(4) for r in range(200) for a in range(6)])
    assert u.min() > 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.02
    assert len(np.unique(u)) > 0.99 * len(u)


def test_uniform_hash_extremes_stay_inside_the_open_interval():
    """u = ((x >> 9) + 0.5) / 2^23 in f32: the largest and smallest hashes must not round to 1.0 / 0.0 (with 24 bits the
    top value did: -log(u) = 0, Gumbel noise +inf, the action forced whatever its probability)."""
    hi = (np.float32(0xFFFFFFFF >> 9) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    lo = (np.float32(0) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    assert hi.dtype == np.float32 and 0.0 < lo and hi < 1.0 and np.isfinite(np.log(-np.log(hi))) and np.isfinite(np.log(-np.log(lo)))
    # and the library agrees with this formula on real hashes
    for row in range(50):
        u = draw_uniform(3, 5, row, 2)
        assert 2.0 ** -24 <= u <= 1.0 - 2.0 ** -24


# ------------------------------------------------------------------------------------------------------ GPU parity
def _run(nets, n_rows, segments, obs, rows_per_tile=128, **kw):
    plan = PolicyPlan(nets, n_rows, segments, rows_per_tile=rows_per_tile)
    dev = nets.device
    out = dict(raw=torch.full((n_rows, 6), float("nan"), device=dev), actions=torch.full((n_rows,), -1, dtype=torch.int32, device=dev),
               action_logp=torch.full((n_rows,), float("nan"), device=dev), logp=torch.full((n_rows, 6), float("nan"), device=dev))
    plan.forward(obs, **out, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("digits", [4, 3])
def test_kernel_matches_reference_actors_and_emulator(digits):
    z, get = cases()
    obs = torch.from_numpy(z["obs"]).cuda()                       # [240, 5, 4, 20]
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[get(f"a{i}.") for i in range(3)], digits=digits)
    order = [0, 1, 2, 2, 0, 1]                                    # six segments of 40 rows
    raw = _run(nets, 240, [(40 * g, 40, k, 0) for g, k in enumerate(order)], obs)["raw"]
    act = _run(nets, 240, [(40 * g, 40, k, TILE_ACT) for g, k in enumerate(order)], obs, deterministic=True)
    tol = TOL if digits == 4 else 2 * TOL
    for g, k in enumerate(order):
        sl = slice(40 * g, 40 * g + 40)
        assert np.abs(log_softmax(raw[sl]) - z[f"a{k}.logits"][sl]).max() < tol + TOL
        assert np.abs(act["logp"][sl] - z[f"a{k}.logits"][sl]).max() < TOL_LOGP
        assert np.array_equal(act["actions"][sl, None], z[f"a{k}.det_action"][sl])
        assert np.abs(act["action_logp"][sl, None] - z[f"a{k}.det_logp"][sl]).max() < TOL_LOGP
        # the emulator replays the same piece products on the same packed image (it rounds each MFMA's 16-term sum once;
        # the matrix pipe's internal order is its own: a few f32 roundings of difference)
        emu = emulate_packed(pack_host(get(f"a{k}."), (5, 4, 20), ACTOR_HEAD, digits), z["obs"][sl].reshape(40, -1))
        assert np.abs(raw[sl] - emu).max() < 1.5e-6
    assert np.isnan(act["raw"]).all()                             # actor tiles do not write raw outputs


@pytest.mark.gpu
def test_kernel_critics_mixed_with_actors_one_launch():
    z, get = cases()
    obs = torch.from_numpy(z["obs"]).cuda()
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[get("a0."), get("a1.")], critics=[get("c0."), get("c1.")])
    assert nets.n_out == [6, 6, 1, 1]
    segs = [(0, 100, 0, TILE_ACT), (100, 140, 1, TILE_ACT), (0, 60, 3, 0), (60, 180, 2, 0)]
    plan = PolicyPlan(nets, 240, segs, rows_per_tile=64)
    values = torch.zeros(240, device="cuda")
    actions = torch.zeros(240, dtype=torch.int32, device="cuda")
    alp = torch.zeros(240, device="cuda")
    plan.forward(obs, raw=values, actions=actions, action_logp=alp, deterministic=True)
    v = values.cpu().numpy()
    assert np.abs(v[:60, None] - z["c1.values"][:60]).max() < TOL and np.abs(v[60:, None] - z["c0.values"][60:]).max() < TOL
    a = actions.cpu().numpy()
    assert np.array_equal(a[:100, None], z["a0.det_action"][:100]) and np.array_equal(a[100:, None], z["a1.det_action"][100:])


@pytest.mark.gpu
def test_kernel_second_grid_and_large_rows():
    """unident_s: 9x5 grid, 21 conv positions, 900-byte rows (115 KB of LDS per workgroup: the opt-in size)."""
    z, get = cases("policy_cases_unident_s.npz")
    obs = torch.from_numpy(z["obs"]).cuda()
    nets = PackedNetworks.from_state_dicts((9, 5, 20), "cuda", actors=[get("a0.")], critics=[get("c0.")])
    r = _run(nets, 96, [(0, 96, 0, TILE_ACT)], obs, deterministic=True)
    assert np.abs(r["logp"] - z["a0.logits"]).max() < TOL_LOGP
    v = _run(nets, 96, [(0, 96, 1, 0)], obs)["raw"].reshape(-1)[:96 * 1]
    # raw of a 1-output network is written with stride n_out = 1
    assert np.abs(v[:96, None] - z["c0.values"]).max() < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(5, 5), (8, 5), (4, 3), (3, 3), (14, 9)])
def test_kernel_other_grids_vs_oracle(W, H):
    """Grids of the other shipped layouts (random0/1 5x5, random3 8x5, corridor 14x9 -- one wavefront per workgroup)
    and the smallest ones, random weights, real-valued int8 inputs in the observation range; ragged row counts."""
    rng = np.random.default_rng(W * 100 + H)
    n = 203
    obs_np = (rng.random((n, W, H, 20)) < 0.15).astype(np.int8) * rng.integers(1, 21, size=(n, W, H, 20), dtype=np.int8)
    actor, critic = random_net(W, H, 6, ACTOR_HEAD, 1), random_net(W, H, 1, CRITIC_HEAD, 2)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[actor], critics=[critic])
    obs = torch.from_numpy(obs_np).cuda()
    want = reference_forward(actor, obs_np, ACTOR_HEAD)
    scale = max(1.0, np.abs(want).max())
    raw = _run(nets, n, [(0, 101, 0, 0), (101, 102, 0, 0)], obs, rows_per_tile=64)["raw"]
    assert np.abs(raw - want).max() < 4e-6 * scale
    v = torch.full((n,), float("nan"), device="cuda")
    PolicyPlan(nets, n, [(3, 197, 1, 0)], rows_per_tile=32).forward(obs, raw=v)
    v = v.cpu().numpy()
    wantv = reference_forward(critic, obs_np, CRITIC_HEAD)[:, 0]
    assert np.isnan(v[:3]).all() and np.isnan(v[200:]).all()
    assert np.abs(v[3:200] - wantv[3:200]).max() < 4e-6 * max(1.0, np.abs(wantv).max())


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,blocks", [(9, 5, 5), (9, 5, 6), (5, 5, 5), (8, 5, 9), (9, 5, 1), (5, 4, 17), (5, 4, 18), (5, 4, 9), (5, 4, 2), (5, 4, 1)])
def test_cooperative_blocks(W, H, blocks, monkeypatch):
    """A stretch of 4 k + 1 (4 k + 2) blocks on a four-wavefront workgroup: the last block (two blocks) is shared by four (two)
    wavefronts that split its conv positions and hand their partial fc1 sums to the group's leader (policy_ops.hip, "cooperative
    blocks").  The 5x4 grid runs eight wavefronts per workgroup: there a stretch of 8 k + 1 (8 k + 2) blocks shares its last block(s)
    among wavefronts 0-3, synchronised by LDS hand-shake words instead of barriers.  Against the numpy restatement of the reference's f32 forward, and against the same launch with every block on one
    wavefront (OC_POLICY_COOP=0): equal to f32 rounding -- and NOT bit for bit, which shows the shared path ran; deterministic actions
    agree wherever the two best log-probabilities are further apart than that rounding.  Ragged last block, actors and critics."""
    from oracle.policy_oracle import log_softmax, reference_forward
    monkeypatch.setenv("OC_POLICY_DUO", "0")                        # (grids of up to 25 tiles otherwise run two blocks per wavefront: no shared blocks there)
    rng = np.random.default_rng(W * 1000 + H * 10 + blocks)
    wgs = 128                                                       # (a table of at least half as many tiles as CUs maps tile i to workgroup i)
    n = wgs * blocks * 32 - 7                                       # every workgroup a stretch of `blocks` blocks, the very last one ragged
    obs_np = (rng.random((n, W, H, 20)) < 0.15).astype(np.int8) * rng.integers(1, 21, size=(n, W, H, 20), dtype=np.int8)
    actor, critic = random_net(W, H, 6, ACTOR_HEAD, 11), random_net(W, H, 1, CRITIC_HEAD, 12)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[actor], critics=[critic])
    obs = torch.from_numpy(obs_np).cuda()
    got = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("OC_POLICY_COOP", coop)
        a = _run(nets, n, [(0, n, 0, TILE_ACT)], obs, rows_per_tile=blocks * 32, deterministic=True)
        v = _run(nets, n, [(0, n, 1, 0)], obs, rows_per_tile=blocks * 32)["raw"].reshape(-1)[:n]
        got[coop] = (a, v)
    want_lp = log_softmax(reference_forward(actor, obs_np, ACTOR_HEAD))
    want_v = reference_forward(critic, obs_np, CRITIC_HEAD).reshape(-1)
    scale = max(1.0, float(np.abs(want_lp).max()), float(np.abs(want_v).max()))
    for coop in ("1", "0"):
        a, v = got[coop]
        assert np.abs(a["logp"] - want_lp).max() < 4e-6 * scale and np.abs(v - want_v).max() < 4e-6 * scale, coop
        top2 = np.sort(want_lp, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4
        assert np.array_equal(a["actions"][clear], want_lp.argmax(1)[clear]) and (a["actions"] >= 0).all()
    (a1, v1), (a0, v0) = got["1"], got["0"]
    assert np.abs(a1["logp"] - a0["logp"]).max() < 2e-6 * scale and np.abs(v1 - v0).max() < 2e-6 * scale
    shared = blocks % 8 == 1 if (W, H) == (5, 4) else blocks % 4 in (1, 2) and (W - 2) * (H - 2) >= 4
    if (W, H) == (5, 4) and not shared:                             # (eight wavefronts: only a stretch's 8 k + 1st block is shared)
        assert np.array_equal(a1["logp"], a0["logp"]) and np.array_equal(v1, v0)
        return
    assert shared and not (np.array_equal(a1["logp"], a0["logp"]) and np.array_equal(v1, v0))
    # only the shared blocks differ: the rows of a workgroup's first 4 k blocks are bit-identical
    rows_wg = blocks * 32
    full = np.concatenate([np.arange(w * rows_wg, w * rows_wg + (blocks - blocks % 4) * 32) for w in range(wgs)])
    full = full[full < n]
    assert np.array_equal(a1["logp"][full], a0["logp"][full]) and np.array_equal(v1[full], v0[full])


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,digits", [(9, 5, 3), (8, 5, 3), (5, 5, 3), (3, 3, 3), (4, 3, 3), (5, 3, 3), (9, 5, 4), (8, 5, 2), (5, 5, 1)])
def test_pipelined_position_loop_equals_the_plain_loop(W, H, digits, monkeypatch):
    """Builds with one wavefront per SIMD (every grid but 5x4) run the position loop pipelined inside the wavefront (policy_mono_loop.h,
    round 6): conv(p+1) and fc1(p-1) beside the conversion of position p, on two sets of conv accumulators.  The same instructions on
    the same values in the same accumulation order as the plain loop (OC_POLICY_MONO=0): BIT for bit, for odd and even numbers of conv
    positions down to one and two (3x3, 4x3), every digit count, ragged last blocks, actors (argmax and sampled) and critics; and
    against the numpy restatement of the reference's f32 forward."""
    from oracle.policy_oracle import log_softmax, reference_forward
    if not native.DIAGNOSTICS:
        pytest.skip("the pipelined position loop is compiled into csrc/liboc_hip_diag.so (OC_DIAGNOSTICS=1) only: DESIGN.md 4.2")
    monkeypatch.setenv("OC_POLICY_COOP", "0")
    monkeypatch.setenv("OC_POLICY_DUO", "0")
    actor, critic = random_net(W, H, 6, ACTOR_HEAD, 31), random_net(W, H, 1, CRITIC_HEAD, 32)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[actor], critics=[critic], digits=digits)
    rng = np.random.default_rng(W * 100 + H * 10 + digits)
    n = 128 * 3 * 32 - 11                                           # 128 workgroups x three blocks, the very last one ragged
    obs_np = (rng.random((n, W, H, 20)) < 0.15).astype(np.int8) * rng.integers(1, 21, size=(n, W, H, 20), dtype=np.int8)
    obs = torch.from_numpy(obs_np).cuda()
    got = {}
    for mono in ("1", "0"):
        monkeypatch.setenv("OC_POLICY_MONO", mono)
        a = _run(nets, n, [(0, n, 0, TILE_ACT)], obs, rows_per_tile=96, deterministic=True)
        smp = _run(nets, n, [(0, n, 0, TILE_ACT)], obs, rows_per_tile=96, deterministic=False, seed=7, counter=3)
        v = _run(nets, n, [(0, n, 1, 0)], obs, rows_per_tile=96)["raw"].reshape(-1)[:n]
        got[mono] = (a, smp, v)
    (a1, s1, v1), (a0, s0, v0) = got["1"], got["0"]
    assert np.array_equal(a1["logp"], a0["logp"]) and np.array_equal(a1["actions"], a0["actions"]) and np.array_equal(a1["action_logp"], a0["action_logp"])
    assert np.array_equal(s1["actions"], s0["actions"]) and np.array_equal(s1["action_logp"], s0["action_logp"]) and np.array_equal(v1, v0)
    want_lp = log_softmax(reference_forward(actor, obs_np, ACTOR_HEAD))
    want_v = reference_forward(critic, obs_np, CRITIC_HEAD).reshape(-1)
    scale = max(1.0, float(np.abs(want_lp).max()), float(np.abs(want_v).max()))
    tol = {1: 2e-2, 2: 2e-4, 3: 4e-6, 4: 4e-6}[digits] * scale      # (one / two conv digits: 8- / 16-bit conv weights)
    assert np.abs(a1["logp"] - want_lp).max() < tol and np.abs(v1 - want_v).max() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(5, 4), (5, 5), (3, 3)])
def test_two_blocks_per_wavefront_equal_the_one_block_kernel(W, H, monkeypatch):
    """oc_policy_duo_kernel (round 6: one wavefront per SIMD interleaving TWO 32-row blocks, policy_duo_kernel.h / policy_duo_body.h)
    against oc_policy_kernel (OC_POLICY_DUO=0) on the same launches: BIT for bit -- the arithmetic and the order of every accumulation
    are the same, only the instruction order differs -- for every way a stretch of 1 .. 19 blocks falls onto the four wavefronts
    (rounds of pairs, a tail of pairs and / or single blocks, ragged last block), actors (sampled and deterministic) and critics, and
    against the numpy restatement of the reference's f32 forward."""
    from oracle.policy_oracle import log_softmax, reference_forward
    monkeypatch.setenv("OC_POLICY_COOP", "0")                       # (the one-block kernel's cooperative blocks sum their fc1 partials in another order)
    actor, critic = random_net(W, H, 6, ACTOR_HEAD, 21), random_net(W, H, 1, CRITIC_HEAD, 22)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[actor], critics=[critic])
    wgs = 128
    for blocks in list(range(1, 12)) + [16, 17, 19]:
        rng = np.random.default_rng(W * 100 + H * 10 + blocks)
        n = wgs * blocks * 32 - 13
        obs_np = (rng.random((n, W, H, 20)) < 0.15).astype(np.int8) * rng.integers(1, 21, size=(n, W, H, 20), dtype=np.int8)
        obs = torch.from_numpy(obs_np).cuda()
        got = {}
        for duo in ("1", "0"):
            monkeypatch.setenv("OC_POLICY_DUO", duo)
            a = _run(nets, n, [(0, n, 0, TILE_ACT)], obs, rows_per_tile=blocks * 32, deterministic=True)
            smp = _run(nets, n, [(0, n, 0, TILE_ACT)], obs, rows_per_tile=blocks * 32, deterministic=False, seed=5, counter=blocks)
            v = _run(nets, n, [(0, n, 1, 0)], obs, rows_per_tile=blocks * 32)["raw"].reshape(-1)[:n]
            got[duo] = (a, smp, v)
        (a1, s1, v1), (a0, s0, v0) = got["1"], got["0"]
        assert np.array_equal(a1["logp"], a0["logp"]) and np.array_equal(a1["actions"], a0["actions"]) and np.array_equal(a1["action_logp"], a0["action_logp"]), blocks
        assert np.array_equal(s1["actions"], s0["actions"]) and np.array_equal(s1["action_logp"], s0["action_logp"]), blocks
        assert np.array_equal(v1, v0), blocks
        if blocks in (1, 5, 17):
            want_lp = log_softmax(reference_forward(actor, obs_np, ACTOR_HEAD))
            want_v = reference_forward(critic, obs_np, CRITIC_HEAD).reshape(-1)
            scale = max(1.0, float(np.abs(want_lp).max()), float(np.abs(want_v).max()))
            assert np.abs(a1["logp"] - want_lp).max() < 4e-6 * scale and np.abs(v1 - want_v).max() < 4e-6 * scale, blocks


@pytest.mark.gpu
def test_multi_grid_launch_equals_per_grid_launches(monkeypatch):
    """oc_policy_forward_multi: three grids (6, 21 and 9 conv positions), actors and critics, ragged segments, ONE launch whose tile
    table is balanced by block cost -- every output identical to the three per-grid launches with the same seeds and counter.
    (Cooperative blocks off: which blocks are shared depends on how a plan cuts the rows, and a shared block sums its fc1 partials
    in another order -- equal to f32 rounding, test_cooperative_blocks, not bit for bit across plans.)"""
    monkeypatch.setenv("OC_POLICY_COOP", "0")
    from diverse_conventions_amd.fused_policy import MultiPolicyPlan, tile_slot
    rng = np.random.default_rng(31)
    grids, rows = [(5, 4), (9, 5), (5, 5)], [1000, 333, 2049]
    nets, segs, obs = [], [], []
    for (W, H), n in zip(grids, rows):
        actors = [random_net(W, H, 6, ACTOR_HEAD, 100 + W * H + i) for i in range(2)]
        critics = [random_net(W, H, 1, CRITIC_HEAD, 200 + W * H)]
        nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=critics))
        cut = n // 3
        segs.append([(0, cut, 0, TILE_ACT), (cut, n - cut, 1, TILE_ACT), (0, n, 2, 0)])
        obs.append(torch.from_numpy(((rng.random((n, W, H, 20)) < 0.15) * rng.integers(1, 21, (n, W, H, 20))).astype(np.int8)).cuda())
    seeds, counter = [11, 22, 33], 5
    new = lambda n, w, dt, fill: torch.full((n, w) if w else (n,), fill, dtype=dt, device="cuda")
    want = []
    for i, n in enumerate(rows):
        o = dict(raw=new(n, 0, torch.float32, float("nan")), actions=new(n, 0, torch.int32, -1), action_logp=new(n, 0, torch.float32, float("nan")),
                 logp=new(n, 6, torch.float32, float("nan")))
        PolicyPlan(nets[i], n, segs[i]).forward(obs[i], seed=seeds[i], counter=counter, **o)
        want.append(o)
    got = [dict(raw=new(n, 0, torch.float32, float("nan")), actions=new(n, 0, torch.int32, -1), action_logp=new(n, 0, torch.float32, float("nan")),
                logp=new(n, 6, torch.float32, float("nan"))) for n in rows]
    assert MultiPolicyPlan(nets, rows, segs, workgroups=64).tiles_per_workgroup == 1      # 9 segments on 64 workgroups: one tile each
    mp = MultiPolicyPlan(nets, rows, segs, workgroups=7)                                  # fewer workgroups than segments: packed stretches
    tl = mp.tiles.cpu().numpy().reshape(mp.workgroups, mp.tiles_per_workgroup, 4)
    live = tl[:, :, 1] > 0
    assert set((tl[:, :, 3][live] >> 16).tolist()) == {0, 1, 2} and tile_slot(2) == 2 << 16
    # every workgroup carries (about) the same cost: blocks weighted by conv positions + 2.5
    cost_of = np.array([(W - 2) * (H - 2) + 2.5 for W, H in grids])
    load = ((tl[:, :, 1] + 31) // 32 * cost_of[tl[:, :, 3] >> 16] * live).sum(1)
    assert load.max() - load.min() <= 2 * cost_of.max() and load.min() > 0
    # and the tiles cover every segment exactly once
    for i, sg in enumerate(segs):
        for row0, cnt, net, flags in sg:
            m = live & ((tl[:, :, 3] >> 16) == i) & (tl[:, :, 2] == net) & ((tl[:, :, 3] & 0xFFFF) == flags)
            spans = sorted((int(r), int(c)) for r, c in zip(tl[:, :, 0][m], tl[:, :, 1][m]) if row0 <= r < row0 + cnt)
            assert spans[0][0] == row0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] == row0 + cnt
    wave_plan = MultiPolicyPlan(nets, rows, segs, waves_per_workgroup=1)                    # the caller's choice: one-wavefront workgroups
    assert wave_plan.waves_per_workgroup == 1 and mp.waves_per_workgroup == 4
    # packed stretches, one tile per workgroup, wave tiles (one-wavefront workgroups, the network in L2)
    for plan in (mp, MultiPolicyPlan(nets, rows, segs), wave_plan):
        for g in got:
            g["raw"].fill_(float("nan")); g["actions"].fill_(-1); g["action_logp"].fill_(float("nan")); g["logp"].fill_(float("nan"))
        plan.forward(obs, raw=[g["raw"] for g in got], actions=[g["actions"] for g in got], action_logp=[g["action_logp"] for g in got],
                     logp=[g["logp"] for g in got], seeds=seeds, counter=counter)
        torch.cuda.synchronize()
        for i in range(3):
            for key in ("raw", "actions", "action_logp", "logp"):
                assert torch.equal(got[i][key], want[i][key]), (i, key, plan.tiles_per_workgroup)
            assert int((got[i]["actions"] < 0).sum()) == 0 and not bool(torch.isnan(got[i]["raw"]).any())


@pytest.mark.gpu
def test_pipelined_position_loop_on_one_two_and_three_positions(monkeypatch):
    """Grids of 3x3, 4x3 and 5x3 tiles (one, two, three conv positions) through a multi-grid launch -- four-wavefront workgroups, conv
    digits in registers: the pipelined position loop's shortest cases (policy_mono_loop.h: no step at all, a last step only, one full
    step) -- against the per-grid launches, which run these small grids on the eight-wavefront build and its plain loop: bit for bit."""
    if not native.DIAGNOSTICS:
        pytest.skip("the pipelined position loop is compiled into csrc/liboc_hip_diag.so (OC_DIAGNOSTICS=1) only: DESIGN.md 4.2")
    monkeypatch.setenv("OC_POLICY_COOP", "0")
    from diverse_conventions_amd.fused_policy import MultiPolicyPlan
    rng = np.random.default_rng(77)
    grids, rows = [(3, 3), (4, 3), (5, 3)], [700, 333, 1029]
    nets, segs, obs = [], [], []
    for (W, H), n in zip(grids, rows):
        nets.append(PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, 300 + W * H)],
                                                    critics=[random_net(W, H, 1, CRITIC_HEAD, 400 + W * H)]))
        segs.append([(0, n, 0, TILE_ACT), (0, n, 1, 0)])
        obs.append(torch.from_numpy(((rng.random((n, W, H, 20)) < 0.15) * rng.integers(1, 21, (n, W, H, 20))).astype(np.int8)).cuda())
    new = lambda n, w, dt, fill: torch.full((n, w) if w else (n,), fill, dtype=dt, device="cuda")
    mk = lambda n: dict(raw=new(n, 0, torch.float32, float("nan")), actions=new(n, 0, torch.int32, -1), action_logp=new(n, 0, torch.float32, float("nan")),
                        logp=new(n, 6, torch.float32, float("nan")))
    want = []
    for i, n in enumerate(rows):
        o = mk(n)
        PolicyPlan(nets[i], n, segs[i]).forward(obs[i], seed=9, counter=2, **o)
        want.append(o)
    for mono in ("1", "0"):
        monkeypatch.setenv("OC_POLICY_MONO", mono)
        got = [mk(n) for n in rows]
        MultiPolicyPlan(nets, rows, segs).forward(obs, raw=[g["raw"] for g in got], actions=[g["actions"] for g in got],
                                                  action_logp=[g["action_logp"] for g in got], logp=[g["logp"] for g in got], seeds=[9, 9, 9], counter=2)
        torch.cuda.synchronize()
        for i in range(3):
            for key in ("raw", "actions", "action_logp", "logp"):
                assert torch.equal(got[i][key], want[i][key]), (mono, i, key)


@pytest.mark.gpu
def test_device_packer_equals_host_packer():
    for (W, H, n_out, head) in [(5, 4, 6, ACTOR_HEAD), (9, 5, 1, CRITIC_HEAD)]:
        sd = random_net(W, H, n_out, head, 11)
        for digits in (1, 4):
            nets = PackedNetworks((W, H, 20), 2, "cuda", digits=digits)
            nets.pack(1, sd, head)
            torch.cuda.synchronize()
            assert np.array_equal(nets.image[1].cpu().numpy(), pack_host(sd, (W, H, 20), head, digits))
            assert not nets.image[0].any()


@pytest.mark.gpu
def test_sampling_is_the_inverse_cdf_of_the_exported_uniform():
    """A sampled action is the first whose cumulative probability exceeds the row's uniform hash(seed, counter, row, 0)
    (Categorical.sample() of the reference, train/MAPPO/utils/distributions.py:20-25, with a reproducible uniform)."""
    z, get = cases()
    obs = torch.from_numpy(z["obs"]).cuda()
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[get("a0.")])
    r = _run(nets, 240, [(0, 240, 0, TILE_ACT)], obs, seed=1234, counter=9)
    u = np.array([draw_uniform(1234, 9, row, 0) for row in range(240)], dtype=np.float64)
    cum = np.cumsum(np.exp(r["logp"].astype(np.float64)), axis=-1)
    cum /= cum[:, -1:]
    expect = np.minimum((cum <= u[:, None]).sum(-1), 5)
    clear = np.abs(cum - u[:, None]).min(-1) > 1e-5               # away from a boundary the draw is reproducible
    assert clear.sum() > 230 and np.array_equal(r["actions"][clear], expect[clear])
    assert np.allclose(r["action_logp"], np.take_along_axis(r["logp"], r["actions"][:, None].astype(np.int64), 1)[:, 0])
    # different counters give different draws; the empirical law follows the probabilities
    big = obs[:1].expand(20000, 5, 4, 20).contiguous()
    rr = _run(nets, 20000, [(0, 20000, 0, TILE_ACT)], big, rows_per_tile=512, seed=5, counter=1)
    freq = np.bincount(rr["actions"], minlength=6) / 20000
    assert np.abs(freq - np.exp(z["a0.logits"][0])).max() < 0.015
    r2 = _run(nets, 240, [(0, 240, 0, TILE_ACT)], obs, seed=1234, counter=10)
    assert (r2["actions"] != r["actions"]).mean() > 0.2


@pytest.mark.gpu
def test_a_sampled_action_always_has_positive_mass():
    """Head biases of -300 on the last three actions: their softmax mass underflows to exactly 0 in f32, so no draw may select them
    however close the row's uniform is to 1.  (The threshold is u x the last value of the very cumulative chain the comparison runs
    over; as u x (own + other half's sum) a threshold rounded up to the chain's end used to fall through to action n_out - 1.)
    Two million draws, every one an action of positive mass, and the law over the three live actions follows their probabilities."""
    actor = random_net(5, 4, 6, ACTOR_HEAD, 9)
    actor[ACTOR_HEAD + ".bias"] = np.array([0.3, -0.2, 0.1, -300.0, -300.0, -300.0], np.float32)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[actor])
    n = 1 << 21
    obs = torch.zeros((n, 5, 4, 20), dtype=torch.int8, device="cuda")
    obs[:, 1, 2, 0] = 1
    plan = PolicyPlan(nets, n, [(0, n, 0, TILE_ACT)])
    acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    logp = torch.zeros((n, 6), device="cuda")
    for counter in (0, 1):
        plan.forward(obs, actions=acts, logp=logp, seed=11, counter=counter)
        torch.cuda.synchronize()
        assert int(acts.min()) >= 0 and int(acts.max()) <= 2
        p = np.exp(logp[0].cpu().numpy())                                 # (f32, as the kernel's own masses: exp(-300) = 0)
        assert p[3:].max() == 0.0
        freq = np.bincount(acts.cpu().numpy(), minlength=6) / n
        assert np.abs(freq - p).max() < 2e-3


@pytest.mark.gpu
def test_activation_range_audit_and_saturation():
    """The dense layers carry hidden activations as f16 pieces at 2^4 x their value: above 65 504 / 16 = 4094 they saturate, where
    the reference's f32 forward does not (documented in PackedNetworks; the reference's trained networks sit at O(1-100)).  (i) NaN /
    Inf parameters are refused at pack time; (ii) `activation_range` reports the f32 maxima and flags the overflow; (iii) a network
    scaled so that its activations reach the thousands but stay under the limit still matches the reference to f32 class (relative to
    its output scale); (iv) over the limit the outputs are finite and deterministic -- the clamp, not inf - inf."""
    from diverse_conventions_amd.fused_policy import ACTIVATION_LIMIT
    rng = np.random.default_rng(4)
    obs_np = ((rng.random((256, 5, 4, 20)) < 0.15) * rng.integers(1, 21, (256, 5, 4, 20))).astype(np.int8)
    obs = torch.from_numpy(obs_np).cuda()
    base = random_net(5, 4, 6, ACTOR_HEAD, 31)
    bad = dict(base)
    bad["base.cnn.cnn.3.weight"] = base["base.cnn.cnn.3.weight"].copy()
    bad["base.cnn.cnn.3.weight"][3, 7] = np.inf
    with pytest.raises(ValueError, match="non-finite"):
        PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[base, bad])          # (one check for everything packed)
    with pytest.raises(ValueError, match="network 0: non-finite values in base.cnn.cnn.3.weight"):
        PackedNetworks((5, 4, 20), 1, "cuda").pack(0, bad, check_finite=True)
    repacked = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[base])
    repacked.pack(0, bad)                                               # a re-pack does not check by itself ...
    with pytest.raises(ValueError, match="non-finite"):
        repacked.check_finite()                                         # ... the batched check does
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[base])
    r0 = nets.activation_range(0, obs)
    assert not r0["saturates"] and r0["limit"] == ACTIVATION_LIMIT == 4094.0
    ref = reference_forward(base, obs_np, ACTOR_HEAD)
    assert abs(r0["out"] - np.abs(ref).max()) < 1e-3 * max(1.0, np.abs(ref).max())
    # (iii) conv weights and bias x s: conv activations x s, everything behind them x s as well (ReLU is positively homogeneous up to
    # the later biases) -- choose s so that the largest hidden activation lands at ~3000
    peak = max(r0["conv"], r0["fc1"], r0["fc2"])
    big = dict(base)
    s = np.float32(3000.0 / peak)
    big["base.cnn.cnn.0.weight"], big["base.cnn.cnn.0.bias"] = base["base.cnn.cnn.0.weight"] * s, base["base.cnn.cnn.0.bias"] * s
    nets_big = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[big])
    rb = nets_big.activation_range(0, obs)
    assert not rb["saturates"] and max(rb["conv"], rb["fc1"], rb["fc2"]) > 1500
    got = _run(nets_big, 256, [(0, 256, 0, 0)], obs)
    want = reference_forward(big, obs_np, ACTOR_HEAD)
    assert np.abs(got["raw"] - want).max() < 4e-6 * max(1.0, np.abs(want).max())
    # (iv) ten times more: over the limit -- flagged by the audit, finite and repeatable in the kernel
    huge = dict(big)
    huge["base.cnn.cnn.0.weight"], huge["base.cnn.cnn.0.bias"] = big["base.cnn.cnn.0.weight"] * 10, big["base.cnn.cnn.0.bias"] * 10
    nets_huge = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[huge])
    assert nets_huge.activation_range(0, obs)["saturates"]
    g1 = _run(nets_huge, 256, [(0, 256, 0, 0)], obs)
    g2 = _run(nets_huge, 256, [(0, 256, 0, 0)], obs)
    assert np.isfinite(g1["raw"]).all() and np.array_equal(g1["raw"], g2["raw"])


@pytest.mark.gpu
def test_bad_tiles_are_clamped_not_faulting():
    z, get = cases()
    obs = torch.from_numpy(z["obs"]).cuda()
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[get("a0.")])
    plan = PolicyPlan(nets, 240, [(0, 240, 0, 0)], rows_per_tile=128)
    plan.tiles = torch.tensor([[200, 500, 7, 0], [-5, 10, 0, 0], [100000, 32, 0, 0], [0, 64, -3, 0]], dtype=torch.int32, device="cuda")
    plan.n_tiles = 4
    raw = torch.full((250, 6), float("nan"), device="cuda")      # 10 canary rows behind the 240 real ones
    plan.forward(obs, raw=raw[:240])
    torch.cuda.synchronize()
    raw = raw.cpu().numpy()
    assert np.isnan(raw[240:]).all() and np.isnan(raw[64:200]).all()
    want = reference_forward(get("a0."), z["obs"], ACTOR_HEAD)
    assert np.abs(raw[200:240] - want[200:240]).max() < TOL and np.abs(raw[:64] - want[:64]).max() < TOL


@pytest.mark.gpu
def test_policy_objects_create_destroy_cycles_do_not_leak():
    """50 cycles of pack (device kernel) / plan / forward / drop: device memory returns to where it was."""
    z, get = cases()
    obs = torch.from_numpy(z["obs"]).cuda()
    sd = get("a0.")
    torch.cuda.synchronize()
    base = None
    for i in range(50):
        nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[sd, sd], critics=[get("c0.")])
        plan = PolicyPlan(nets, 240, [(0, 240, i % 2, TILE_ACT), (0, 240, 2, 0)])
        a = torch.zeros(240, dtype=torch.int32, device="cuda")
        v = torch.zeros(240, device="cuda")
        plan.forward(obs, raw=v, actions=a, deterministic=True)
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy()[:, None], z["a0.det_action"])
        del nets, plan, a, v
        if i == 4:
            base = torch.cuda.memory_allocated()
    assert torch.cuda.memory_allocated() <= base


@pytest.mark.gpu
def test_kernel_two_million_rows_sampled_against_the_oracle():
    """Full-size address arithmetic: 2 097 152 rows (2^20 kitchens x 2 sides, 839 MB of observations) in one launch,
    ragged segment sizes, a sample of rows from every part of the tensor against the numpy restatement."""
    rng = np.random.default_rng(21)
    n = 1 << 21
    actor, critic = random_net(5, 4, 6, ACTOR_HEAD, 5), random_net(5, 4, 1, CRITIC_HEAD, 6)
    nets = PackedNetworks.from_state_dicts((5, 4, 20), "cuda", actors=[actor], critics=[critic])
    g = torch.Generator(device="cuda").manual_seed(3)
    obs = (torch.rand((n, 5, 4, 20), device="cuda", generator=g) < 0.12).to(torch.int8) * \
        torch.randint(1, 21, (n, 5, 4, 20), device="cuda", generator=g, dtype=torch.int8)
    cuts = [0, 777_777, 1_000_001, n]
    segs = [(cuts[i], cuts[i + 1] - cuts[i], 0, TILE_ACT) for i in range(3)] + [(cuts[i], cuts[i + 1] - cuts[i], 1, 0) for i in range(3)]
    plan = PolicyPlan(nets, n, segs)
    acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    logp = torch.full((n, 6), float("nan"), device="cuda")
    vals = torch.full((n,), float("nan"), device="cuda")
    plan.forward(obs, raw=vals, actions=acts, logp=logp, deterministic=True)
    torch.cuda.synchronize()
    assert int((acts < 0).sum()) == 0 and not bool(torch.isnan(vals).any()) and not bool(torch.isnan(logp).any())
    pick = np.unique(np.concatenate([rng.integers(0, n, 600), np.arange(0, 40), np.arange(n - 40, n),
                                     np.arange(777_777 - 40, 777_777 + 40), np.arange(1_000_001 - 40, 1_000_001 + 40)]))
    o = obs[torch.from_numpy(pick).cuda()].cpu().numpy()
    # random networks of this scale put logits at +-12: 1.6e-5 is 1.3e-6 of that, the size of the f32 restatement's own
    # distance from float64 arithmetic on these inputs (tools/policy_accuracy.py, profiles/r02_policy_accuracy.log)
    want = log_softmax(reference_forward(actor, o, ACTOR_HEAD))
    assert np.abs(logp[torch.from_numpy(pick).cuda()].cpu().numpy() - want).max() < 1.6e-5
    assert np.abs(vals[torch.from_numpy(pick).cuda()].cpu().numpy() - reference_forward(critic, o, CRITIC_HEAD)[:, 0]).max() < 1.6e-5


@pytest.mark.gpu
def test_the_binding_shown_in_integration_md_runs():
    """INTEGRATION.md 2a is code a maintainer would paste into the reference: execute that block as written."""
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n# train/XD/fused_policies\.py\n(.*?)```", text, re.S).group(1)
    block = block.replace('"diverse-conventions_amd/csrc/liboc_hip.so"', repr(native.LIB_PATH))
    ns = {}
    exec(compile(block, "INTEGRATION.md:2a", "exec"), ns)
    z, get = cases()

    class Module:                                        # stands for an R_Actor / R_Critic: only state_dict() is used
        def __init__(self, sd):
            self.sd = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in sd.items()}

        def state_dict(self):
            return self.sd
    fp = ns["FusedPolicies"](5, 4, [Module(get("a0.")), Module(get("a1.")), Module(get("c0."))],
                             [ACTOR_HEAD, ACTOR_HEAD, CRITIC_HEAD], "cuda")
    obs = torch.from_numpy(z["obs"]).cuda()
    tiles = fp.plan([(0, 120, 0, 1), (120, 120, 1, 1), (0, 240, 2, 0)])
    actions = torch.zeros(240, dtype=torch.int32, device="cuda")
    logp, values = torch.zeros(240, device="cuda"), torch.zeros(240, device="cuda")
    fp.act(obs, tiles, actions, logp, values, step=0)
    torch.cuda.synchronize()
    assert np.abs(values.cpu().numpy()[:, None] - z["c0.values"]).max() < TOL
    a = actions.cpu().numpy().astype(np.int64)
    ref = np.concatenate([z["a0.logits"][:120], z["a1.logits"][120:]])
    assert np.abs(logp.cpu().numpy() - np.take_along_axis(ref, a[:, None], 1)[:, 0]).max() < TOL_LOGP


# ------------------------------------------------------------------------------------------- gathered rows (mixed play)
@pytest.mark.gpu
def test_partition_rows_is_a_stable_two_way_partition():
    from diverse_conventions_amd.fused_policy import partition_rows
    rng = np.random.default_rng(2)
    for lists, length, period in [(6, 1, 2), (4, 63, 2), (10, 1024, 2), (7, 1025, 1), (4, 5000, 4), (2, 65471, 2)]:
        sel = (rng.random((lists, length)) < rng.random((lists, 1))).astype(np.int32) * rng.integers(1, 9, (lists, length)).astype(np.int32)
        sel[0] = 0
        if lists > 1:
            sel[1] = 5
        index, count = partition_rows(torch.from_numpy(sel).cuda(), id_period=period)
        index, count = index.cpu().numpy(), count.cpu().numpy()
        for l in range(lists):
            ids = (l % period) * length + np.arange(length)
            zero = ids[sel[l] == 0]
            assert count[l, 0] == len(zero) and count[l, 1] == length - len(zero)
            assert np.array_equal(index[l, :len(zero)], zero) and np.array_equal(index[l, len(zero):], ids[sel[l] != 0])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n", [((5, 4), 1000), ((9, 5), 333), ((5, 5), 517), ((8, 5), 97)])
def test_gather_tiles_evaluate_exactly_the_rows_of_their_class(shape, n):
    """Two actors and two critics over 2 n rows; every row belongs to policy 0 or 1 by a random choice per (side, kitchen).  The
    gathered launch (each network reads only its rows through the index lists) must give bit-for-bit what the select launch gives
    (both candidates laid over all rows, each writing only its own): same staging bytes, same arithmetic, same draws -- on row
    sizes that are whole 16-byte pieces (5x4, 8x5) and on ragged ones (9x5: 900 B, 5x5: 500 B) whose last piece runs into the next
    row, with the tensor's very last row in play and canary bytes behind the observation tensor."""
    from diverse_conventions_amd.fused_policy import GatherPlan, partition_rows, tile_select
    W, H = shape
    rb = W * H * 20
    rng = np.random.default_rng(W * 31 + H)
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=[random_net(W, H, 6, ACTOR_HEAD, 1), random_net(W, H, 6, ACTOR_HEAD, 2)],
                                           critics=[random_net(W, H, 1, CRITIC_HEAD, 3), random_net(W, H, 1, CRITIC_HEAD, 4)])
    rows = 2 * n
    backing = torch.full((rows * rb + 64,), 0x55, dtype=torch.int8, device="cuda")
    obs = backing[:rows * rb].view(rows, W, H, 20)
    obs.copy_(torch.from_numpy(((rng.random((rows, W, H, 20)) < 0.15) * rng.integers(1, 21, (rows, W, H, 20))).astype(np.int8)))
    for trial, p1 in enumerate((0.5, 0.0, 1.0, 0.03)):
        choice = (rng.random((2, n)) < p1).astype(np.int32)
        if trial == 0:
            choice[1, -1] = 1                                          # the tensor's last row belongs to the second policy ...
        if trial == 3:
            choice[1, -1] = 0                                          # ... or to the first
        dchoice = torch.from_numpy(choice).cuda()
        outs = []
        for mode in ("select", "gather"):
            out = dict(raw=torch.full((rows, 1), float("nan"), device="cuda"), actions=torch.full((rows,), -1, dtype=torch.int32, device="cuda"),
                       action_logp=torch.full((rows,), float("nan"), device="cuda"), logp=torch.full((rows, 6), float("nan"), device="cuda"))
            if mode == "select":
                segs = []
                for side in range(2):
                    for c in range(2):
                        segs += [(side * n, n, c, TILE_ACT | tile_select(c)), (side * n, n, 2 + c, tile_select(c))]
                PolicyPlan(nets, rows, segs).forward(obs, **out, seed=9, counter=4, row_select=dchoice.view(-1))
            else:
                tiles = []
                for side in range(2):
                    for c in range(2):
                        tiles += [(side, c, c, TILE_ACT), (side, c, 2 + c, 0)]
                index, count = partition_rows(dchoice, id_period=2)
                assert int(count[:, 1].sum()) == int(choice.sum())
                GatherPlan(nets, rows, n, tiles).forward(obs, index, count, **out, seed=9, counter=4)
            torch.cuda.synchronize()
            outs.append({k: v.cpu().numpy() for k, v in out.items()})
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), (trial, k)
        assert not np.isnan(outs[1]["raw"]).any() and (outs[1]["actions"] >= 0).all()
        assert bool((backing[rows * rb:] == 0x55).all())


# ------------------------------------------------------------------------------------------- one launch per training step
@pytest.mark.gpu
@pytest.mark.parametrize("layout,threads,k", [("simple", 37, 2), ("simple", 820, 2), ("unident_s", 45, 1), ("random1", 33, 3), ("random3", 64, 1)])
def test_env_step_folded_into_the_policy_launch_equals_two_launches(layout, threads, k, layout_params, monkeypatch):
    """oc_policy_rollout with the env step folded into the policy launch (OC_POLICY_FOLD=1: every actor block bumps the arrival
    counters of the kitchen groups it touches, the wavefront that completes a group steps it inside the same launch) against the
    two-launch path (OC_POLICY_FOLD=0), same seeds: actions, log-probs, values, observations, rewards, dones and the engines' final
    state records must be identical -- time-major collection (fresh slots) and in-place stepping, pairing segments that do not
    line up with the 32-kitchen fold groups, batches with one block per wavefront and with several, critics in the table.
    (Cooperative blocks off for the two-launch side: the folded build has none, and a shared block rounds its fc1 sum differently.)
    The folded step is a measured-and-lost experiment: it exists in the diagnostics build of the library only, so this
    test runs under `OC_DIAGNOSTICS=1 python -m pytest -m gpu tests/test_fused_policy.py -k folded` and is skipped on the product build."""
    if not native.DIAGNOSTICS:
        pytest.skip("the folded env step lives in csrc/liboc_hip_diag.so (OC_DIAGNOSTICS=1) only")
    monkeypatch.setenv("OC_POLICY_COOP", "0")
    from diverse_conventions_amd import FusedPairedRollout, OvercookedEngine, TrajectoryBuffer
    params = layout_params[layout]
    W, H = params["width"], params["height"]
    G = 2 * k + 1
    n, T = threads * G, 11
    actors = [random_net(W, H, 6, ACTOR_HEAD, 10 + i) for i in range(k + 1)]
    critics = [random_net(W, H, 1, CRITIC_HEAD, 20 + i) for i in range(2)]
    nets = PackedNetworks.from_state_dicts((W, H, 20), "cuda", actors=actors, critics=critics)
    critic_of = [[k + 1 + (g % 2) for g in range(G)], [k + 1 + ((g + 1) % 2) for g in range(G)]]
    got = {}
    for fold in ("0", "1"):
        monkeypatch.setenv("OC_POLICY_FOLD", fold)
        eng = OvercookedEngine(params, n, 0)
        eng.stagger(40)                                              # episode ends (horizon resets) inside the test's steps
        eng.episode_control(timesteps=(torch.arange(n, device="cuda") % 7 + 190).to(torch.int32))
        ro = FusedPairedRollout(eng, nets, threads, k, critic_nets=critic_of, seed=3)
        tr = TrajectoryBuffer(eng, T)
        lp, va = torch.zeros((T, 2, n), device="cuda"), torch.zeros((T, 2, n), device="cuda")
        before = native.lib().oc_debug_fold_launches()
        ro.collect(tr, logp=lp, values=va)
        obs, rew, done = ro.step(6)                                  # in place, continuing the draw counter
        torch.cuda.synchronize()
        # the path the variable asks for really ran: the T collecting steps fold; the six in-place steps with critics do not (a critic block
        # that comes behind its group's env step would read rows already overwritten: they run as two launches, round 4) -- same results
        assert native.lib().oc_debug_fold_launches() - before == (T if fold == "1" else 0)
        got[fold] = dict(tobs=tr.obs.clone(), tact=tr.actions.clone(), trew=tr.reward.clone(), tdone=tr.done.clone(), lp=lp, va=va,
                         obs=obs.clone(), rew=rew.clone(), done=done.clone(), act=ro.actions.clone(), state=eng.export_states().clone())
        assert int(tr.done.sum()) > 0
        eng.close()
    for key in got["0"]:
        assert torch.equal(got["0"][key], got["1"][key]), key
