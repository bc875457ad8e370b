"""numpy restatement of the reference's update-side data path.  TEST INFRASTRUCTURE ONLY.

* normalize_advantages: train/XD/xd.py:240-244 (= train/MAPPO/r_mappo.py:177-181).
* feed_forward_batches / partial_ff_batches: SharedReplayBuffer.feed_forward_generator / partial_ff_generator
  (train/MAPPO/utils/shared_buffer.py:306-366, 368-429) for a given permutation.
* diag_insert / part_insert: SharedReplayBuffer.diaginsert / partinsert (train/MAPPO/utils/shared_buffer.py:150-220).

Arrays are in the reference's layout: [T or T+1, N, A, ...].
Parity: pinned by tests/golden/minibatch_cases.npz (outputs of the reference itself).
"""
import numpy as np

f32 = np.float32


def normalize_advantages(adv, active_masks=None, std_over="reference"):
    """adv [T, N, A, 1] f32, active_masks [T, N, A, 1] (= buffer.active_masks[:-1]) or None.
    -> (normalised advantages f32, mean, std).

    The reference masks inactive entries with NaN, takes nanmean for the mean, and then selects the entries for
    the std with `copy != nan` -- true everywhere, NaN included -- so its std (and with it every advantage) is
    NaN as soon as one entry is inactive (xd.py:243).  std_over="reference" keeps that; "active" is the std over
    the active entries only."""
    adv = np.asarray(adv, dtype=f32)
    act = np.ones(adv.shape, dtype=bool) if active_masks is None else (np.asarray(active_masks) != 0)
    x = adv[act].astype(np.float64)
    mean = f32(x.mean()) if x.size else f32(np.nan)
    if std_over == "reference" and not act.all():
        std = f32(np.nan)
    else:
        std = f32(x.std(ddof=1)) if x.size > 1 else f32(np.nan)                 # torch .std() is unbiased
    return ((adv - mean) / (std + f32(1e-5))).astype(f32), mean, std


def _rows(a, rows):
    return a.reshape(-1, *a.shape[3:])[rows]


def feed_forward_batches(buf, advantages, perm, num_mini_batch, mini_batch_size=None):
    """buf: dict of reference-layout arrays (obs, rnn_states, rnn_states_critic, actions, value_preds, returns,
    masks, active_masks, action_log_probs[, available_actions]).  Yields dicts of minibatch arrays."""
    T, N, A = buf["actions"].shape[:3]
    mbs = mini_batch_size or (T * N * A) // num_mini_batch                       # :316-326
    for i in range(num_mini_batch):                                             # :329
        idx = perm[i * mbs:(i + 1) * mbs]
        out = {k: _rows(buf[k][:T], idx) for k in ("obs", "rnn_states", "rnn_states_critic", "value_preds", "returns",
                                                   "masks", "active_masks") if k in buf}          # [:-1] views, :331-343
        out.update({k: _rows(buf[k], idx) for k in ("actions", "action_log_probs")})
        out["adv_targ"] = _rows(advantages, idx)
        if "available_actions" in buf:
            out["available_actions"] = _rows(buf["available_actions"][:T], idx)
        yield out


def partial_ff_batches(buf, advantages, agent, perm, num_mini_batch, mini_batch_size=None):
    """One agent's rows only: row r of the flat batch is (t, n) = divmod(r, N) (:393-405)."""
    T, N = buf["actions"].shape[:2]
    mbs = mini_batch_size or (T * N) // num_mini_batch
    sel = lambda a: a[:T, :, agent].reshape(T * N, *a.shape[3:])
    for i in range(num_mini_batch):
        idx = perm[i * mbs:(i + 1) * mbs]
        out = {k: sel(buf[k])[idx] for k in ("obs", "rnn_states", "rnn_states_critic", "value_preds", "returns", "masks",
                                             "active_masks", "actions", "action_log_probs") if k in buf}
        out["adv_targ"] = sel(advantages)[idx]
        if "available_actions" in buf:
            out["available_actions"] = sel(buf["available_actions"])[idx]
        yield out


def diag_insert(dst, src, step):
    """dst [R, N, ...] <- src [step + 1, ...] along the diagonal the reference's diaginsert picks at `step`
    (:167-168): with d = N - step - 1, dst[i, i + d] = src[i] for every i the diagonal holds.  Returns the
    number of rows written."""
    R, N = dst.shape[:2]
    d = N - step - 1
    n = min(R, N - d) if d >= 0 else min(R + d, N)
    for i in range(n):
        if d >= 0:
            dst[i, i + d] = src[i]
        else:
            dst[i - d, i] = src[i]
    return n


def part_insert(dst, src, step):
    """partinsert (:200-202): dst[step, :step] = src."""
    dst[step, :step] = src
