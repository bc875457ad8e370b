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


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32_cuda(name, t):
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous float32 CUDA tensor")


_workspaces = {}


def normalize_advantages(advantages, active_masks=None, std_over="reference", out=None):
    """`(adv - nanmean) / (std + 1e-5)` of the reference's `calc_advantanges` / `R_MAPPO.train`
    (train/XD/xd.py:240-244, train/MAPPO/r_mappo.py:177-181) in three launches, no host synchronisation.

    advantages [T, N, A, 1] f32 (e.g. from `gae_returns`); active_masks = `buffer.active_masks[:-1]` or None (all active).
    std_over="reference": the reference selects the entries for the std with `copy != nan`, which keeps the NaN-masked
    inactive ones, so std and advantages are NaN whenever an entry is inactive; "active": std of the active entries.
    -> (normalised advantages (`out`, may be `advantages` itself), stats f32[2] = {mean, std} on the device)
    """
    if std_over not in ("reference", "active"):
        raise ValueError("std_over must be 'reference' or 'active'")
    _f32_cuda("advantages", advantages)
    if active_masks is not None:
        active_masks = active_masks.contiguous()
        _f32_cuda("active_masks", active_masks)
        if active_masks.numel() != advantages.numel():
            raise ValueError("active_masks must have the shape of advantages (pass buffer.active_masks[:-1])")
    if out is None:
        out = torch.empty_like(advantages)
    _f32_cuda("out", out)
    L = native.lib()
    ws = _workspaces.get(advantages.device)
    if ws is None:
        ws = _workspaces[advantages.device] = torch.empty(L.oc_advantage_workspace_bytes(), dtype=torch.uint8, device=advantages.device)
    stats = torch.empty(2, dtype=torch.float32, device=advantages.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    native.check(L.oc_advantage_normalize(p(advantages), p(active_masks), advantages.numel(), 1 if std_over == "active" else 0,
                                          p(out), p(stats), p(ws), _stream(advantages)))
    return out, stats


def gather_rows(src, index, out=None, dtype=None):
    """out[i] = src[index[i]] for byte rows: src int8 [R, ...] (contiguous, trailing sizes multiply to a multiple of 4),
    index int64 [n] on the device (None = all rows in order); dtype torch.int8 (default) or torch.float32 (cast fused).
    Indices outside [0, R) are clamped, not faulted on."""
    if src.dtype != torch.int8 or not src.is_cuda or not src.is_contiguous():
        raise ValueError("src must be a contiguous int8 CUDA tensor")
    dtype = dtype or torch.int8
    if dtype not in (torch.int8, torch.float32):
        raise ValueError("dtype must be torch.int8 or torch.float32")
    R = src.shape[0]
    row_bytes = src[0].numel() if R else 0
    if index is not None and (index.dtype != torch.int64 or not index.is_cuda or not index.is_contiguous() or index.dim() != 1):
        raise ValueError("index must be a contiguous 1-d int64 CUDA tensor")
    n = R if index is None else index.numel()
    if out is None:
        out = torch.empty((n, *src.shape[1:]), dtype=dtype, device=src.device)
    elif out.dtype != dtype or not out.is_contiguous() or out.numel() != n * row_bytes or out.device != src.device:
        raise ValueError("out must be a contiguous tensor of n rows in the requested dtype on src's device")
    if n == 0:
        return out
    native.check(native.lib().oc_gather_rows(ctypes.c_void_p(src.data_ptr()), R, row_bytes,
                                             ctypes.c_void_p(index.data_ptr()) if index is not None else None, n,
                                             ctypes.c_void_p(out.data_ptr()), 1 if dtype == torch.float32 else 0, _stream(src)))
    return out


def gather_columns(columns, index, row_mul=1, row_add=0, norm_stats=None, outs=None):
    """Gather up to 8 four-byte-element columns in one launch: outs[c][i] = columns[c][index[i] * row_mul + row_add].
    columns: contiguous CUDA tensors [R_c, ...] of a 4-byte dtype; norm_stats: {column position: stats f32[2]} applies
    `(x - mean) / (std + 1e-5)` on the fly (stats from `normalize_advantages`)."""
    n_cols = len(columns)
    if not 1 <= n_cols <= 8:
        raise ValueError("1..8 columns per call")
    if index.dtype != torch.int64 or not index.is_cuda or not index.is_contiguous() or index.dim() != 1:
        raise ValueError("index must be a contiguous 1-d int64 CUDA tensor")
    n = index.numel()
    vp = ctypes.c_void_p
    src, dst, stats = (vp * n_cols)(), (vp * n_cols)(), (vp * n_cols)()
    rows, words = (ctypes.c_int64 * n_cols)(), (ctypes.c_int32 * n_cols)()
    results = []
    for c, t in enumerate(columns):
        if t.element_size() != 4 or not t.is_cuda or not t.is_contiguous() or t.dim() < 1:
            raise ValueError("columns must be contiguous CUDA tensors of a 4-byte dtype")
        w = t[0].numel() if t.shape[0] else 0
        if w < 1:
            raise ValueError("empty column")
        o = outs[c] if outs is not None else torch.empty((n, *t.shape[1:]), dtype=t.dtype, device=t.device)
        if o.dtype != t.dtype or not o.is_contiguous() or o.numel() != n * w:
            raise ValueError("bad output tensor for column %d" % c)
        src[c], dst[c], rows[c], words[c] = t.data_ptr(), o.data_ptr(), t.shape[0], w
        s = (norm_stats or {}).get(c)
        if s is not None:
            _f32_cuda("norm_stats", s)
            if t.dtype != torch.float32:
                raise ValueError("only float32 columns can be normalised")
        stats[c] = s.data_ptr() if s is not None else None
        results.append(o)
    if n:
        native.check(native.lib().oc_gather_columns(n_cols, src, rows, words, dst, stats, vp(index.data_ptr()), n, int(row_mul),
                                                    int(row_add), _stream(index)))
    return results


def pack_columns(columns, n_rows):
    """Interleave up to 8 one-word columns (contiguous 4-byte CUDA tensors with at least n_rows elements) into
    sample records f32 [n_rows, 8]: one 32-byte sector per sample for `gather_records`."""
    n_cols = len(columns)
    if not 1 <= n_cols <= 8:
        raise ValueError("1..8 columns per record")
    vp = ctypes.c_void_p
    src = (vp * n_cols)()
    for c, t in enumerate(columns):
        if t.element_size() != 4 or not t.is_cuda or not t.is_contiguous() or t.numel() < n_rows:
            raise ValueError("columns must be contiguous 4-byte CUDA tensors of at least n_rows elements")
        src[c] = t.data_ptr()
    packed = torch.empty((n_rows, 8), dtype=torch.float32, device=columns[0].device)
    native.check(native.lib().oc_pack_columns(n_cols, src, n_rows, vp(packed.data_ptr()), _stream(packed)))
    return packed


def gather_records(packed, n_cols, index, row_mul=1, row_add=0, norm_col=-1, norm_stats=None, dtypes=None):
    """-> n_cols tensors [n, 1]: column c of records packed[index * row_mul + row_add]; `norm_col` is normalised with
    `norm_stats` = {mean, std} while it is gathered."""
    if packed.dtype != torch.float32 or packed.dim() != 2 or packed.shape[1] != 8 or not packed.is_contiguous():
        raise ValueError("packed must come from pack_columns")
    if index.dtype != torch.int64 or not index.is_cuda or not index.is_contiguous() or index.dim() != 1:
        raise ValueError("index must be a contiguous 1-d int64 CUDA tensor")
    n = index.numel()
    vp = ctypes.c_void_p
    outs = [torch.empty((n, 1), dtype=(dtypes[c] if dtypes else torch.float32), device=packed.device) for c in range(n_cols)]
    dst = (vp * n_cols)(*[o.data_ptr() for o in outs])
    if norm_col >= 0:
        _f32_cuda("norm_stats", norm_stats)
    if n:
        native.check(native.lib().oc_gather_records(vp(packed.data_ptr()), packed.shape[0], n_cols, dst, norm_col,
                                                    vp(norm_stats.data_ptr()) if norm_col >= 0 else None, vp(index.data_ptr()), n,
                                                    int(row_mul), int(row_add), _stream(packed)))
    return outs
