"""Time-major rollout storage that the step kernel fills directly.

The reference runner copies every observation three times after the env step -- `vobs.obs.clone()`, the
implicit int8->float32 cast into `turn_obs`, and `SharedReplayBuffer.chooseinsert` into the float32
`(T+1, N, agents, W, H, C)` buffer (train/MAPPO/main_player.py:245,229; utils/shared_buffer.py:115-148):
more HBM traffic than the env kernel itself, at 4x the bytes.  Here the kernel's destination for step t IS
slot t+1 of an int8 trajectory tensor; nothing is copied and the dtype stays int8 until a consumer casts.
"""
import torch

from .engine import OvercookedEngine


def _numel(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


def wire_layout(T, P, n, rec_words):
    """Word offsets of the compact trajectory's four fields inside ONE flat int32 buffer, in wire order:
    [records (T+1)·n·rec/4 | actions T·P·n | reward T·P·n | done T·n], every field starting on a 16-byte boundary (the encode kernel
    reads records as uint4; the step kernel writes them so) -> ({name: (offset, shape)}, total words, a multiple of 4).
    4S+16 bytes of records + 4P + 4P + 4 bytes per kitchen-step instead of P·S·(5P+10) bytes of observations."""
    fields, off = {}, 0
    for name, shape in (("records", (T + 1, n, rec_words)), ("actions", (T, P, n)), ("reward", (T, P, n)), ("done", (T, n))):
        words = 1
        for d in shape:
            words *= int(d)
        fields[name] = (off, tuple(int(d) for d in shape))
        off += words + (-words) % 4
    return fields, off


class TrajectoryBuffer:
    """obs int8 [T+1, P, N, W, H, C]; actions int32 [T, P, N]; reward int32 [T, P, N]; done int32 [T, N];
    optionally the compact native state records int32 [T+1, N, rec/4] (for `TrajectoryExchange`).

    With `store_states=True` the compact fields -- records, actions, reward, done -- are views of ONE flat int32 buffer, `wire`, laid
    out in the order the exchange sends them (`wire_layout`): `TrajectoryExchange.gather_buffer` hands that buffer to the one
    all-gather as it is (no concatenation pass), and the step launches themselves write slot t+1 of the records (oc_step's
    d_records: no export pass either).  `obs_slots=1` keeps a single observation slot that every step overwrites in place (a rank
    that only needs the compact trajectory, e.g. the exchange leg of bench.py)."""

    def __init__(self, engine: OvercookedEngine, length: int, store_states: bool = False, obs_slots=None):
        self.engine, self.T = engine, int(length)
        n, dev = engine.num_kitchens, engine.device
        P = self.P = engine.num_players
        self.obs_slots = self.T + 1 if obs_slots is None else int(obs_slots)
        if self.obs_slots not in (1, self.T + 1):
            raise ValueError("obs_slots is T + 1 (time-major, the default) or 1 (overwritten in place)")
        self.obs = torch.zeros((self.obs_slots, P, n, engine.width, engine.height, engine.channels), dtype=torch.int8, device=dev)
        self.wire = None
        if store_states:
            fields, words = wire_layout(self.T, P, n, engine.record_bytes // 4)
            self.wire = torch.zeros(words, dtype=torch.int32, device=dev)
            view = lambda name: self.wire[fields[name][0]:fields[name][0] + _numel(fields[name][1])].view(fields[name][1])
            self.records, self.actions, self.reward, self.done = view("records"), view("actions"), view("reward"), view("done")
        else:
            self.actions = torch.zeros((self.T, P, n), dtype=torch.int32, device=dev)
            self.reward = torch.zeros((self.T, P, n), dtype=torch.int32, device=dev)
            self.done = torch.zeros((self.T, n), dtype=torch.int32, device=dev)
            self.records = None
        self.t = 0
        self.start()

    def _obs_slot(self, t):
        return self.obs[t if self.obs_slots > 1 else 0]

    def require_time_major_obs(self, who):
        """Callers that hand `obs` to a native rollout which writes slot t+1 at step t (T+1 slots behind the pointer): a buffer
        with `obs_slots=1` holds one slot, and the rollout would write T slots past its end."""
        slot = self.obs[0].numel()
        if self.obs_slots != self.T + 1 or self.obs.numel() != (self.T + 1) * slot or not self.obs.is_contiguous():
            raise ValueError(f"{who} fills time-major observation slots: the trajectory needs obs_slots = T + 1 "
                             f"(this one has {self.obs_slots} for T = {self.T})")

    def start(self):
        """Slot 0 := observation of the engine's current states (episodes keep running, like n_reset)."""
        self.t = 0
        self.engine.observe(self.obs[0])
        if self.records is not None:
            self.engine.export_states(out=self.records[0])
        return self.obs[0]

    def step(self, actions):
        """actions [P, N] or [P, N, 1] (any integer dtype) -> (obs[t+1], reward[t], done[t]) views.  ONE launch: the step kernel
        writes the observation slot, reward, done and -- with store_states -- the post-step records of slot t+1."""
        if self.t >= self.T:
            raise IndexError("trajectory is full: call after_update() or start()")
        t = self.t
        self.actions[t].copy_(actions.reshape(self.P, -1))
        self.engine.step(self.actions[t], obs=self._obs_slot(t + 1), reward=self.reward[t], done=self.done[t],
                         records=self.records[t + 1] if self.records is not None else None)
        self.t = t + 1
        return self._obs_slot(t + 1), self.reward[t], self.done[t]

    def rollout_tape(self, actions):
        """Fill the whole buffer from a pre-drawn action tape [T, P, N], one launch per step, all enqueued from C (oc_rollout);
        records included when the buffer stores them."""
        self.start()
        self.actions.copy_(actions.reshape(self.T, self.P, -1))
        self.engine.rollout(self.actions, self.T, obs=self.obs[1:] if self.obs_slots > 1 else self.obs[0], reward=self.reward, done=self.done,
                            records=self.records[1:] if self.records is not None else None)
        self.t = self.T
        return self

    def rollout(self, actions):
        """Fill the whole buffer from a pre-drawn action tape [T, P, N] in ONE launch (oc_rollout_fused: the
        kitchens' state stays on chip between steps; 9.5 vs 13.6 us per step at 65 536 kitchens)."""
        self.require_time_major_obs("rollout()")
        self.start()
        self.actions.copy_(actions.reshape(self.T, self.P, -1))
        self.engine.rollout_fused(self.T, actions=self.actions, obs=self.obs[1:], reward=self.reward, done=self.done,
                                  records=self.records[1:] if self.records is not None else None)
        self.t = self.T
        return self

    def after_update(self):
        """Last slot becomes the first of the next segment (reference shared_buffer.py:222-231)."""
        if self.obs_slots > 1:
            self.obs[0].copy_(self.obs[self.T])
        if self.records is not None:
            self.records[0].copy_(self.records[self.T])
        self.t = 0

    def episode_returns(self):
        """Mean return of the episodes that ENDED inside the buffer, per kitchen -> (mean f32 [N], count int64 [N]); NaN where
        none ended.  What the reference's runners accumulate as `running_score` and push into `scores` at every done
        (train/XD/xd_player.py:224-230); feed it to `sharding.segment_scores`.  Assumes slot 0 starts an episode."""
        return episode_returns_of(self.reward[:self.t, 0], self.done[:self.t])      # the shared reward (sum over players)

    def agent_view(self, agent: int):
        """Observations of one player as the reference buffer indexes them: [T+1, N, W, H, C] (int8 view)."""
        return self.obs[:, agent]


def episode_returns_of(reward, done):
    """reward [T, N], done [T, N] (any numeric dtype) -> (mean return of the episodes that ended inside the T steps f32 [N], their
    count int64 [N]); NaN where none ended.  The reference adds every step's reward to `running_score`, pushes it into the segment's
    score list at a done and zeroes it (train/XD/xd_player.py:217-230): the sum of those pushes is the reward total up to the last
    done.  Pinned to the reference's own lists by tests/test_pairing_golden.py.  Rewards must not be negative (Overcooked's are not)."""
    r = reward.to(torch.float32)
    d = done.to(torch.float32)
    upto_last_done = (r.cumsum(0) * d).amax(0)
    count = d.sum(0)
    return upto_last_done / count, count.to(torch.int64)
