"""Runner-side ops next to the env step, thin torch front-ends of the C ABI."""
import ctypes

import torch

from . import native


def gae_returns(rewards, value_preds, masks, bad_masks=None, gamma=0.99, gae_lambda=0.95, value_normalizer=None,
                use_proper_time_limits=False, returns=None, advantages=True):
    """One-launch GAE over a trajectory batch: what `SharedReplayBuffer.compute_returns(next_value, vn)`
    (reference train/MAPPO/utils/shared_buffer.py:248-304, use_gae=True) leaves in `buffer.returns[:-1]`, plus
    `returns[:-1] - denormalize(value_preds[:-1])` (r_mappo.py:174-177) when `advantages` is true.

    rewards [T, N, A, 1]; value_preds / masks / bad_masks [T+1, N, A, 1] f32 on the GPU, `value_preds[-1]` already
    holding next_value.  `value_normalizer`: object with `running_mean_var()` (the reference's ValueNorm/PopArt)
    or a (mean, std) pair, or None.  -> (returns [T+1, N, A, 1] with rows 0..T-1 filled, advantages [T, N, A, 1] | None)
    """
    T = rewards.shape[0]
    for t in (rewards, value_preds, masks):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError("gae_returns wants contiguous float32 CUDA tensors")
    M = rewards[0].numel()
    if value_preds.shape[0] != T + 1 or masks.shape[0] != T + 1 or value_preds[0].numel() != M:
        raise ValueError("value_preds and masks must be [T+1, ...] with the trailing shape of rewards")
    mean, std, denorm = 0.0, 1.0, 0
    if value_normalizer is not None:
        if hasattr(value_normalizer, "running_mean_var"):
            m, v = value_normalizer.running_mean_var()
            mean, std = float(m.reshape(-1)[0]), float(torch.sqrt(v).reshape(-1)[0])
        else:
            mean, std = (float(x) for x in value_normalizer)
        denorm = 1
    if returns is None:
        returns = torch.zeros_like(value_preds)
    adv = torch.empty_like(rewards) if advantages else None
    if use_proper_time_limits and bad_masks is None:
        raise ValueError("use_proper_time_limits needs bad_masks")
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    stream = ctypes.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
    native.check(native.lib().oc_gae_returns(p(rewards), p(value_preds), p(masks), p(bad_masks), p(returns), p(adv), T, M,
                                             float(gamma), float(gae_lambda), denorm, mean, std,
                                             1 if use_proper_time_limits else 0, stream))
    return returns, adv
