"""PPO minibatch generators on the device.

The reference's `SharedReplayBuffer.feed_forward_generator` / `partial_ff_generator`
(train/MAPPO/utils/shared_buffer.py:306-366, 368-429) draw `torch.randperm` on the host, flatten every buffer to
`[T*N*A, ...]` and fancy-index ten tensors per minibatch -- the float32 observations twice (`share_obs`, `obs`).
Here the permutation is drawn on the device, the int8 observation rows are gathered straight out of the
trajectory the step kernel wrote (`TrajectoryBuffer.obs`, `[T+1, P, N, W, H, C]`) with the cast to the
consumer's dtype fused, and all scalar columns go in one launch.  For the same permutation the minibatches are
element-for-element the reference's (tests/golden/minibatch_cases.npz).
"""
import torch

from .rollout_ops import gather_columns, gather_records, gather_rows, pack_columns

_SCALARS = ("actions", "value_preds", "returns", "masks", "active_masks", "action_log_probs")


class MinibatchSource:
    """Holds what an update reads.

    obs             int8 [T+1, P, N, W, H, C]  (engine layout: agent-major, as TrajectoryBuffer stores it)
    actions, action_log_probs  [T, N, A, 1];  value_preds, returns, masks, active_masks  [T+1, N, A, 1]
    rnn_states, rnn_states_critic  [T+1, N, A, recurrent_N, hidden] (optional);  available_actions [T+1, N, A, n] (optional;
    Overcooked's are all ones -- sim.cpp:771 -- and a broadcast view is yielded when absent)
    all float32 in the reference's layout, on the device of `obs`.
    """

    def __init__(self, obs, actions, value_preds, returns, masks, active_masks, action_log_probs, rnn_states=None,
                 rnn_states_critic=None, available_actions=None, num_actions=6):
        if obs.dtype != torch.int8 or obs.dim() != 6 or not obs.is_contiguous():
            raise ValueError("obs must be a contiguous int8 [T+1, P, N, W, H, C] tensor")
        self.obs = obs
        self.T, self.A, self.N = obs.shape[0] - 1, obs.shape[1], obs.shape[2]
        self.cols = dict(actions=actions, value_preds=value_preds, returns=returns, masks=masks, active_masks=active_masks,
                         action_log_probs=action_log_probs, rnn_states=rnn_states, rnn_states_critic=rnn_states_critic,
                         available_actions=available_actions)
        for k, t in self.cols.items():
            if t is None:
                continue
            want_t = self.T if k in ("actions", "action_log_probs") else self.T + 1
            if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape[:3]) != (want_t, self.N, self.A):
                raise ValueError(f"{k} must be contiguous float32 [{want_t}, {self.N}, {self.A}, ...]")
        self.num_actions = num_actions

    @classmethod
    def from_trajectory(cls, traj, value_preds, returns, masks, action_log_probs, active_masks=None, **kw):
        """`traj`: a filled TrajectoryBuffer; its int32 `[T, P, N]` actions become the reference's float `[T, N, A, 1]`."""
        actions = traj.actions.permute(0, 2, 1).unsqueeze(-1).float().contiguous()
        if active_masks is None:
            active_masks = torch.ones_like(masks)
        return cls(traj.obs, actions, value_preds, returns, masks, active_masks, action_log_probs, **kw)

    def permutation(self, batch_size, generator=None):
        return torch.randperm(batch_size, device=self.obs.device, generator=generator)

    def _sampler(self, batch_size, num_mini_batch, mini_batch_size, perm, generator):
        if num_mini_batch is None:
            raise ValueError("num_mini_batch is required")
        if mini_batch_size is None:
            if batch_size < num_mini_batch:                                       # shared_buffer.py:316-324
                raise AssertionError(f"PPO requires the batch ({batch_size}) to be at least the number of minibatches ({num_mini_batch})")
            mini_batch_size = batch_size // num_mini_batch
        if perm is None:
            perm = self.permutation(batch_size, generator)
        else:
            perm = torch.as_tensor(perm, dtype=torch.int64, device=self.obs.device).contiguous()
            if perm.numel() and (int(perm.min()) < 0 or int(perm.max()) >= batch_size):
                raise IndexError("permutation entries must lie in [0, batch_size)")
        return [perm[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(num_mini_batch)]   # :326-327

    def _records(self, advantages):
        """Sample records of the six scalar columns (+ advantages): packed afresh at the start of every generator call
        (one streaming pass).  Nothing is cached across calls: the columns are written in place through raw pointers by
        this library's own kernels (gae_returns, normalize_advantages, the policy kernels), which torch's version
        counters do not see, and the caching allocator hands the same address out again for a new tensor."""
        if advantages is not None and (advantages.dtype != torch.float32 or not advantages.is_contiguous()
                                       or advantages.numel() != self.T * self.N * self.A):
            raise ValueError("advantages must be contiguous float32 [T, N, A, 1]")
        cols = [self.cols[k] for k in _SCALARS] + ([advantages] if advantages is not None else [])
        return pack_columns(cols, self.T * self.N * self.A)

    def _batch(self, packed, idx, obs_rows, advantages, row_mul, row_add, obs_dtype, norm_stats):
        obs_batch = gather_rows(self.obs.view(-1, *self.obs.shape[3:]), obs_rows, dtype=obs_dtype)
        n_cols = len(_SCALARS) + (advantages is not None)
        vals = gather_records(packed, n_cols, idx, row_mul, row_add,
                              norm_col=len(_SCALARS) if (advantages is not None and norm_stats is not None) else -1,
                              norm_stats=norm_stats)
        out = dict(zip(_SCALARS + ("adv_targ",), vals))
        wide = [k for k in ("rnn_states", "rnn_states_critic", "available_actions") if self.cols[k] is not None]
        if wide:
            out.update(zip(wide, gather_columns([self.cols[k].view(-1, *self.cols[k].shape[3:]) for k in wide], idx, row_mul, row_add)))
        avail = out.get("available_actions")
        if avail is None:
            avail = torch.ones((1, self.num_actions), dtype=torch.float32, device=self.obs.device).expand(idx.numel(), -1)
        return (obs_batch, obs_batch, out.get("rnn_states"), out.get("rnn_states_critic"), out["actions"], out["value_preds"],
                out["returns"], out["masks"], out["active_masks"], out["action_log_probs"], out.get("adv_targ"), avail)

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, perm=None, generator=None,
                               obs_dtype=torch.float32, norm_stats=None):
        """Yields the reference's 12-tuple (share_obs, obs, rnn_states, rnn_states_critic, actions, value_preds, returns,
        masks, active_masks, old_action_log_probs, adv_targ, available_actions); share_obs IS obs (Overcooked's state is
        its observation, overcooked2_env.py:97-98).  Flat row r = (t*N + n)*A + a (shared_buffer.py:331-343).
        advantages [T, N, A, 1] f32 or None; with `norm_stats` (from normalize_advantages(..., out=None) statistics) raw
        advantages are normalised while they are gathered.  obs_dtype torch.float32 (reference) or torch.int8."""
        T, N, A = self.T, self.N, self.A
        packed = self._records(advantages)
        for idx in self._sampler(T * N * A, num_mini_batch, mini_batch_size, perm, generator):
            a = idx % A
            tn = idx // A
            obs_rows = ((tn // N) * A + a) * N + tn % N                           # (t, a, n) in the agent-major trajectory
            yield self._batch(packed, idx.contiguous(), obs_rows, advantages, 1, 0, obs_dtype, norm_stats)

    def partial_ff_generator(self, advantages, agent, num_mini_batch=None, mini_batch_size=None, perm=None, generator=None,
                             obs_dtype=torch.float32, norm_stats=None):
        """One agent's samples only (shared_buffer.py:368-429): flat row r = t*N + n."""
        T, N, A = self.T, self.N, self.A
        if not 0 <= agent < A:
            raise IndexError("agent index out of range")
        packed = self._records(advantages)
        for idx in self._sampler(T * N, num_mini_batch, mini_batch_size, perm, generator):
            obs_rows = ((idx // N) * A + agent) * N + idx % N
            yield self._batch(packed, idx.contiguous(), obs_rows, advantages, A, agent, obs_dtype, norm_stats)
