"""Update-side data path: advantage normalisation and the minibatch generators.

Golden: tests/golden/minibatch_cases.npz, produced by the reference's XD.calc_advantanges and
SharedReplayBuffer.feed_forward_generator / partial_ff_generator (tests/golden/make_golden_minibatch.py).
Gathers are bit-exact.  Normalised advantages: the reference reduces in f32 (torch), oracle and kernel in f64;
tolerance 2e-6 absolute on mean and std and 1e-5 on the normalised values (unit scale), written below."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import minibatch_oracle as mo

FIELDS = ("obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns", "masks", "active_masks",
          "action_log_probs", "adv_targ", "available_actions")
TUPLE_POS = dict(obs=1, rnn_states=2, rnn_states_critic=3, actions=4, value_preds=5, returns=6, masks=7, active_masks=8,
                 action_log_probs=9, adv_targ=10, available_actions=11)


def cases():
    z = np.load(os.path.join(GOLDEN, "minibatch_cases.npz"))
    for i in range(int(z["n_cases"])):
        yield i, {k[len(f"c{i}_"):]: z[k] for k in z.files if k.startswith(f"c{i}_")}


CASES = list(cases())
ids = lambda x: f"case{x}" if isinstance(x, int) else ""


def gen_adv(c):
    return c.get("generator_advantages", c["advantages"])


def buf_of(c):
    b = {k: c[k] for k in ("obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns", "masks", "active_masks",
                           "action_log_probs")}
    b["available_actions"] = np.ones((*c["masks"].shape[:3], 6), np.float32)
    return b


@pytest.mark.parametrize("i,c", CASES, ids=ids)
def test_oracle_normalisation_matches_reference(i, c):
    out, mean, std = mo.normalize_advantages(c["raw_advantages"], c["active_masks"][:-1])
    if np.isnan(c["advantages"]).all():                       # inactive entries: the reference's std is NaN (xd.py:243)
        assert np.isnan(std) and np.isnan(out).all()
        out, mean, std = mo.normalize_advantages(c["raw_advantages"], c["active_masks"][:-1], std_over="active")
        act = c["active_masks"][:-1] != 0
        x = c["raw_advantages"][act].astype(np.float64)
        assert abs(mean - x.mean()) < 2e-6 and abs(std - x.std(ddof=1)) < 2e-6 and np.isfinite(out).all()
    else:
        np.testing.assert_allclose(out, c["advantages"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("i,c", CASES, ids=ids)
def test_oracle_generators_match_reference(i, c):
    nmb = int(c["num_mini_batch"])
    for name, gen in (("ff", mo.feed_forward_batches(buf_of(c), gen_adv(c), c["ff_perm"], nmb)),
                      ("p0", mo.partial_ff_batches(buf_of(c), gen_adv(c), 0, c["p0_perm"], nmb)),
                      ("p1", mo.partial_ff_batches(buf_of(c), gen_adv(c), 1, c["p1_perm"], nmb))):
        batches = list(gen)
        assert len(batches) == nmb
        for k, b in enumerate(batches):
            for f in FIELDS:
                assert np.array_equal(b[f], c[f"{name}_{f}"][k]), (name, k, f)


def source_of(c, with_available=True):
    import torch
    from diverse_conventions_amd import MinibatchSource
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    obs = t(c["obs"]).permute(0, 2, 1, 3, 4, 5).contiguous()                    # reference [T+1,N,A,...] -> engine [T+1,A,N,...]
    avail = torch.ones((*c["masks"].shape[:3], 6), device="cuda") if with_available else None
    return MinibatchSource(obs, t(c["actions"]), t(c["value_preds"]), t(c["returns"]), t(c["masks"]), t(c["active_masks"]),
                           t(c["action_log_probs"]), t(c["rnn_states"]), t(c["rnn_states_critic"]), avail)


@pytest.mark.gpu
@pytest.mark.parametrize("i,c", CASES, ids=ids)
@pytest.mark.parametrize("obs_dtype", ["float32", "int8"])
def test_kernel_generators_match_reference(i, c, obs_dtype):
    import torch
    src = source_of(c, with_available=(i != 1))
    adv = torch.from_numpy(gen_adv(c)).cuda()
    nmb = int(c["num_mini_batch"])
    dt = getattr(torch, obs_dtype)
    for name, gen in (("ff", src.feed_forward_generator(adv, nmb, perm=c["ff_perm"], obs_dtype=dt)),
                      ("p0", src.partial_ff_generator(adv, 0, nmb, perm=c["p0_perm"], obs_dtype=dt)),
                      ("p1", src.partial_ff_generator(adv, 1, nmb, perm=c["p1_perm"], obs_dtype=dt))):
        batches = list(gen)
        assert len(batches) == nmb
        for k, b in enumerate(batches):
            assert b[0] is b[1] and b[1].dtype == dt
            for f in FIELDS:
                got = b[TUPLE_POS[f]].cpu().numpy()
                assert np.array_equal(got.astype(c[f"{name}_{f}"].dtype), c[f"{name}_{f}"][k]), (name, k, f)


@pytest.mark.gpu
@pytest.mark.parametrize("i,c", CASES, ids=ids)
def test_kernel_normalisation_matches_reference(i, c):
    import torch
    from diverse_conventions_amd import normalize_advantages
    raw = torch.from_numpy(c["raw_advantages"]).cuda()
    act = torch.from_numpy(np.ascontiguousarray(c["active_masks"][:-1])).cuda()
    out, stats = normalize_advantages(raw, act)
    if np.isnan(c["advantages"]).all():
        assert torch.isnan(out).all() and torch.isnan(stats[1]) and torch.isfinite(stats[0])
        out, stats = normalize_advantages(raw, act, std_over="active")
        ref, mean, std = mo.normalize_advantages(c["raw_advantages"], c["active_masks"][:-1], std_over="active")
        assert abs(float(stats[0]) - mean) < 2e-6 and abs(float(stats[1]) - std) < 2e-6
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-5)
    else:
        np.testing.assert_allclose(out.cpu().numpy(), c["advantages"], rtol=0, atol=1e-5)
        out2, _ = normalize_advantages(raw, None)                               # all active: the mask is optional
        assert torch.equal(out, out2)
        out3, _ = normalize_advantages(raw.clone(), act, out=None)
        assert torch.equal(out, out3)                                           # fixed summation order: reproducible


@pytest.mark.gpu
def test_normalise_while_gathering_equals_normalise_then_gather():
    import torch
    from diverse_conventions_amd import normalize_advantages
    i, c = CASES[0]
    src = source_of(c)
    raw = torch.from_numpy(c["raw_advantages"]).cuda()
    normed, stats = normalize_advantages(raw)
    a = list(src.feed_forward_generator(normed, 3, perm=c["ff_perm"]))
    b = list(src.feed_forward_generator(raw, 3, perm=c["ff_perm"], norm_stats=stats))
    for x, y in zip(a, b):
        assert torch.equal(x[10], y[10])


@pytest.mark.gpu
def test_device_permutation_covers_every_sample_once():
    import torch
    i, c = CASES[1]
    src = source_of(c)
    T, N, A = int(c["T"]), int(c["N"]), int(c["A"])
    tag = torch.arange(T * N * A, dtype=torch.float32, device="cuda").view(T, N, A, 1)      # advantages carry the row id
    g = torch.Generator(device="cuda").manual_seed(5)
    seen = torch.cat([b[10].flatten() for b in src.feed_forward_generator(tag, 5, generator=g)]).long()
    assert seen.numel() == T * N * A and torch.equal(seen.sort().values, torch.arange(T * N * A, device="cuda"))
    assert not torch.equal(seen, torch.arange(T * N * A, device="cuda"))
    # every field of a sample comes from the same (t, n, a)
    g = torch.Generator(device="cuda").manual_seed(6)
    for b in src.feed_forward_generator(tag, 4, generator=g, obs_dtype=torch.int8):
        r = b[10].flatten().long().cpu().numpy()
        t, n, a = r // (N * A), (r // A) % N, r % A
        assert np.array_equal(b[1].cpu().numpy(), c["obs"][t, n, a])
        assert np.array_equal(b[5].cpu().numpy(), c["value_preds"][t, n, a])
        assert np.array_equal(b[4].cpu().numpy(), c["actions"][t, n, a])
    with pytest.raises(IndexError):
        list(src.feed_forward_generator(tag, 2, perm=np.array([0, T * N * A])))
    with pytest.raises(AssertionError):
        list(src.feed_forward_generator(tag, T * N * A + 1))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(5, 5, 20), (5, 4, 20), (9, 5, 20), (3, 1, 4), (16, 16, 20)])
@pytest.mark.parametrize("n_rows", [1, 63, 64, 65, 1000, 4097])
def test_gather_rows_vs_numpy(shape, n_rows):
    """Row lengths that are / are not multiples of 16 bytes (500, 400, 900, 12, 5120 B), batches around the 64-row groups."""
    import torch
    from diverse_conventions_amd import gather_rows
    rng = np.random.default_rng(n_rows)
    R = 777
    src = rng.integers(-128, 128, size=(R, *shape), dtype=np.int8)
    idx = rng.integers(0, R, size=n_rows)
    s, ix = torch.from_numpy(src).cuda(), torch.from_numpy(idx).cuda()
    assert np.array_equal(gather_rows(s, ix).cpu().numpy(), src[idx])
    f = gather_rows(s, ix, dtype=torch.float32)
    assert f.dtype == torch.float32 and np.array_equal(f.cpu().numpy(), src[idx].astype(np.float32))
    if n_rows == 1000:
        assert np.array_equal(gather_rows(s, None).cpu().numpy(), src)
        bad = torch.tensor([-5, R + 3, 2], device="cuda")
        assert np.array_equal(gather_rows(s, bad).cpu().numpy(), src[[0, R - 1, 2]])          # clamped, no fault


@pytest.mark.gpu
def test_large_normalisation_vs_float64():
    """26 M advantages (T=200 x 65 536 kitchens x 2): statistics against numpy float64 on the same data."""
    import torch
    from diverse_conventions_amd import normalize_advantages
    g = torch.Generator(device="cuda").manual_seed(1)
    adv = torch.randn((200, 65536, 2, 1), device="cuda", generator=g) * 3.0 + 0.7
    out, stats = normalize_advantages(adv)
    x = adv.cpu().numpy().astype(np.float64)
    assert abs(float(stats[0]) - x.mean()) < 2e-6 and abs(float(stats[1]) - x.std(ddof=1)) < 2e-6
    ref = ((adv.cpu().numpy() - np.float32(x.mean())) / (np.float32(x.std(ddof=1)) + np.float32(1e-5)))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_sample_records_follow_in_place_updates():
    """The packed scalar records are a cache: a buffer written in place between two epochs must show up."""
    import torch
    i, c = CASES[0]
    src = source_of(c)
    adv = torch.from_numpy(gen_adv(c)).cuda()
    first = [b[5].clone() for b in src.feed_forward_generator(adv, 3, perm=c["ff_perm"])]
    src.cols["value_preds"].add_(1.0)
    adv.mul_(2.0)
    second = list(src.feed_forward_generator(adv, 3, perm=c["ff_perm"]))
    for k, (a, b) in enumerate(zip(first, second)):
        assert torch.equal(a + 1.0, b[5])
        assert np.array_equal(b[10].cpu().numpy(), c["ff_adv_targ"][k] * 2.0)


@pytest.mark.gpu
def test_minibatches_see_writes_made_through_raw_pointers():
    """This library's own kernels write their outputs through raw device pointers (no torch version bump): a persistent
    `returns` buffer refilled by gae_returns and advantages normalised in place between two generator calls must be what
    the second call yields -- nothing may be cached across calls."""
    import torch
    from diverse_conventions_amd import gae_returns, normalize_advantages
    i, c = CASES[0]
    src = source_of(c)
    T, N, A = src.T, src.N, src.A
    g = torch.Generator(device="cuda").manual_seed(3)
    adv = torch.randn((T, N, A, 1), device="cuda", generator=g)
    first = [(b[6].clone(), b[10].clone()) for b in src.feed_forward_generator(adv, 3, perm=c["ff_perm"])]
    # same tensors, same data_ptr, same _version: refill `returns` and normalise `adv` in place with the native kernels
    rewards = torch.randn((T, N, A, 1), device="cuda", generator=g)
    v0 = src.cols["returns"]._version, adv._version
    gae_returns(rewards, src.cols["value_preds"], src.cols["masks"], returns=src.cols["returns"])
    normalize_advantages(adv, out=adv)
    assert (src.cols["returns"]._version, adv._version) == v0, "the premise of this test: raw-pointer writes do not bump _version"
    second = list(src.feed_forward_generator(adv, 3, perm=c["ff_perm"]))
    perm = torch.as_tensor(c["ff_perm"]).cuda()
    mb = (T * N * A) // 3
    ret_flat, adv_flat = src.cols["returns"][:-1].reshape(-1, 1), adv.reshape(-1, 1)
    changed = False
    for k, b in enumerate(second):
        idx = perm[k * mb:(k + 1) * mb]
        assert torch.equal(b[6], ret_flat[idx]) and torch.equal(b[10], adv_flat[idx])
        changed |= not torch.equal(b[6], first[k][0])
    assert changed
