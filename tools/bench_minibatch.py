#!/usr/bin/env python3
"""Update-side data path timing at the BASELINE trajectory size (T=200 x 65 536 kitchens x 2 agents, cramped_room
observations): advantage normalisation and one PPO minibatch of obs rows + scalar columns, next to the torch
formulation the reference uses (fancy indexing of flattened buffers, shared_buffer.py:331-360)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diverse_conventions_amd import MinibatchSource, gather_rows, normalize_advantages

T, N, A, W, H, C = 200, 65536, 2, 5, 4, 20
NUM_MINI_BATCH = 8
dev = "cuda"


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


obs = torch.randint(0, 2, (T + 1, A, N, W, H, C), dtype=torch.int8, device=dev)
col = lambda t: torch.randn((t, N, A, 1), device=dev)
src = MinibatchSource(obs, col(T), col(T + 1), col(T + 1), col(T + 1), torch.ones((T + 1, N, A, 1), device=dev), col(T))
adv = col(T)
out = []

# advantage normalisation
ms = timed(lambda: normalize_advantages(adv, src.cols["active_masks"][:-1].contiguous() if False else None))
n = adv.numel()
def torch_norm():
    c = adv.clone(); m = torch.nanmean(c); s = c.flatten().std(); return (adv - m) / (s + 1e-5)
out.append({"op": "normalize_advantages", "elements": n, "ms": round(ms, 4), "algorithmic_GBps": round(n * 12 / ms / 1e6, 1),
            "roofline_frac": round(n * 12 / ms / 1e6 / 8000, 3), "torch_ms": round(timed(torch_norm), 4)})

# one minibatch: obs rows
batch = T * N * A
mbs = batch // NUM_MINI_BATCH
perm = torch.randperm(batch, device=dev)
idx = perm[:mbs].contiguous()
a, tn = idx % A, idx // A
rows = (((tn // N) * A + a) * N + tn % N).contiguous()
flat = obs.view(-1, W, H, C)
row_bytes = W * H * C
for dt, mult in ((torch.int8, 1), (torch.float32, 4)):
    dst = torch.empty((mbs, W, H, C), dtype=dt, device=dev)
    ms = timed(lambda: gather_rows(flat, rows, out=dst, dtype=dt))
    alg = mbs * (row_bytes * (1 + mult) + 8)
    ref = (lambda: flat[rows]) if dt == torch.int8 else (lambda: flat[rows].float())
    out.append({"op": f"gather_rows int8->{str(dt).split('.')[1]}", "rows": mbs, "row_bytes": row_bytes, "ms": round(ms, 4),
                "algorithmic_GBps": round(alg / ms / 1e6, 1), "roofline_frac": round(alg / ms / 1e6 / 8000, 3),
                "torch_index_ms": round(timed(ref), 4), "equal_to_torch": bool(torch.equal(dst, ref()))})
    del dst

# one full minibatch through the generator (obs f32 + 7 scalar columns, advantages normalised on the fly)
_, stats = normalize_advantages(adv, out=torch.empty_like(adv))
def one_epoch(dt):
    for b in src.feed_forward_generator(adv, NUM_MINI_BATCH, perm=perm, obs_dtype=dt, norm_stats=stats):
        pass
for dt in (torch.int8, torch.float32):
    ms = timed(lambda: one_epoch(dt), reps=2)
    out.append({"op": f"feed_forward_generator epoch, obs {str(dt).split('.')[1]}", "samples": batch, "minibatches": NUM_MINI_BATCH,
                "ms": round(ms, 3), "M_samples_per_s": round(batch / ms / 1e3, 1)})

# the reference's formulation on the same GPU, float32 observation buffer in its layout (one minibatch, scaled to an epoch)
del obs, src, flat
torch.cuda.empty_cache()
fobs = torch.zeros((T, N, A, W, H, C), dtype=torch.float32, device=dev).view(-1, W, H, C)       # 42 GB
cols = [torch.randn((T * N * A, 1), device=dev) for _ in range(7)]
def ref_minibatch():
    i = perm[:mbs]
    return [fobs[i], fobs[i]] + [c[i] for c in cols]
ms = timed(ref_minibatch, reps=2)
out.append({"op": "reference formulation (float32 buffers, fancy indexing), one minibatch x %d" % NUM_MINI_BATCH,
            "ms": round(ms * NUM_MINI_BATCH, 3)})
for o in out:
    print(json.dumps(o))
