"""Segment-batched policy inference against outputs of the reference's R_Actor
(tests/golden/policy_cases.npz, from tests/golden/make_golden_policy.py).  Floating point: logits within
2e-5 absolute of the reference's float32 forward (GEMM summation order differs), deterministic actions equal."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from torch_policies import StackedActors, StackedCritics
from diverse_conventions_amd.sharding import segment_policy_indices
from diverse_conventions_amd.sharding import pairing_segments

TOL = 2e-5


def load():
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    sds = [{k[len(f"a{i}."):]: z[k] for k in z.files if k.startswith(f"a{i}.") and "logits" not in k and "det_" not in k}
           for i in range(3)]
    return z, sds


def check(device):
    z, sds = load()
    actors = StackedActors(sds, (5, 4, 20), device)
    obs = torch.from_numpy(z["obs"]).to(device)                       # int8 [240, 5, 4, 20]
    # six segments of 40 kitchens, policies 0,1,2,2,0,1: every actor sees different slices in one call
    order = [0, 1, 2, 2, 0, 1]
    seg = obs.reshape(6, 40, 5, 4, 20)
    logp = actors.log_probs(seg, torch.tensor(order))
    a, lp = actors.act(seg, torch.tensor(order), deterministic=True)
    for g, k in enumerate(order):
        want = z[f"a{k}.logits"][g * 40:(g + 1) * 40]
        assert np.abs(logp[g].cpu().numpy() - want).max() < TOL
        assert np.array_equal(a[g].cpu().numpy(), z[f"a{k}.det_action"][g * 40:(g + 1) * 40])
        assert np.abs(lp[g].cpu().numpy() - z[f"a{k}.det_logp"][g * 40:(g + 1) * 40]).max() < TOL
    # sampling follows the distribution (chi-square-free sanity: empirical frequencies near probabilities)
    big = seg[:1, :1].expand(1, 20000, 5, 4, 20)
    s, _ = actors.act(big, torch.tensor([0]))
    freq = torch.bincount(s.flatten().cpu(), minlength=6).float() / 20000
    assert (freq - logp[0, 0].exp().cpu()).abs().max() < 0.02


def test_stacked_critics_match_reference_cpu():
    z = np.load(os.path.join(GOLDEN, "policy_cases.npz"))
    sds = [{k[len(f"c{i}."):]: z[k] for k in z.files if k.startswith(f"c{i}.") and not k.endswith(".values")} for i in range(2)]
    critics = StackedCritics(sds, (5, 4, 20), "cpu")
    seg = torch.from_numpy(z["obs"]).reshape(4, 60, 5, 4, 20)
    order = [1, 0, 0, 1]
    v = critics.values(seg, torch.tensor(order))
    assert v.shape == (4, 60, 1)
    for g, k in enumerate(order):
        assert np.abs(v[g].numpy() - z[f"c{k}.values"][g * 60:(g + 1) * 60]).max() < TOL


def test_stacked_actors_match_reference_cpu():
    check("cpu")


@pytest.mark.gpu
def test_stacked_actors_match_reference_gpu():
    check("cuda")


def test_segment_policy_lists_match_reference():
    segs = pairing_segments(50, 3)
    assert segment_policy_indices(segs, "ego") == [0, 0, 0, 0, 1, 2, 3]
    assert segment_policy_indices(segs, "partner") == [0, 1, 2, 3, 0, 0, 0]
