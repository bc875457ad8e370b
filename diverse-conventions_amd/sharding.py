"""Kitchens across the GPUs of one node: sharding, convention-pairing segments, trajectory exchange.

The reference runs everything on one GPU (`gpu_id=0`, train/env_utils.py:26) and has no collective
code; this module is the new multi-GPU capability of the north star.  Kitchens never interact, so a
rollout needs NO communication: every rank owns a contiguous block of kitchens and keeps the full
pairing-segment structure at 1/world size.  The only exchange is once per PPO update, when the
cross-play / mixed-play loss wants the whole batch: ranks all-gather *compact* trajectories
(native state records + actions + rewards + dones: 4S+16 B per kitchen-step instead of 40S B of
observations) over RCCL/xGMI and re-featurise peers' shards locally with the encode kernel.
"""
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class KitchenShard:
    """Contiguous block [first, first+count) of `total` kitchens owned by `rank` of `world`."""
    total: int
    world: int
    rank: int

    @property
    def count(self) -> int:
        return self.total // self.world + (1 if self.rank < self.total % self.world else 0)

    @property
    def first(self) -> int:
        base, extra = divmod(self.total, self.world)
        return self.rank * base + min(self.rank, extra)

    def counts(self) -> List[int]:
        return [KitchenShard(self.total, self.world, r).count for r in range(self.world)]


@dataclass(frozen=True)
class Segment:
    """Kitchens [start, stop) of one env object played by (ego policy, partner policy).
    policy index 0 = the convention being trained, i >= 1 = frozen convention i-1."""
    name: str
    start: int
    stop: int
    ego_policy: int
    partner_policy: int


def pairing_segments(threads: int, n_conventions: int) -> List[Segment]:
    """Segment map of the reference's CoMeDi env for the (k+1)-th convention, k = n_conventions
    already trained: `threads*(2k+1)` kitchens laid out [SP | XP0_0..XP0_{k-1} | XP1_0..XP1_{k-1}]
    (train/XD/serial.py:27); ego side plays [new]*(k+1) + [conv_i] (xd_player.py:94), partner side
    [new] + [conv_i] + [new]*k (xd_player.py:76-77)."""
    k = n_conventions
    segs = [Segment("SP", 0, threads, 0, 0)]
    for i in range(k):
        segs.append(Segment(f"XP0_{i}", threads * (1 + i), threads * (2 + i), 0, 1 + i))
    for i in range(k):
        segs.append(Segment(f"XP1_{i}", threads * (1 + k + i), threads * (2 + k + i), 1 + i, 0))
    return segs


def aligned_threads(kitchens: int, n_conventions: int, multiple: int = 32) -> int:
    """Kitchens per pairing segment for a budget of `kitchens` on one GPU: the largest multiple of 32 (the fused policy kernel's
    block of rows) that fits the 2k+1 segments.  A segment of 264 kitchens is 8.25 blocks for each of its networks -- the ragged ninth
    costs a whole block and pushes a workgroup into one more round; 256 are exactly 8.  Measured with actors and critics in the loop
    (tools/exp_aligned_segments.py): config 4's per-GPU share 51.0 us per step at 31 x 264 kitchens, 41.5 at 31 x 256 (+19 % agent-env
    steps/s); config 5's per-layout share 30.0 us at 63 x 130, 24.6 at 63 x 128.  Segments shorter than `multiple` are left alone."""
    per = int(kitchens) // (2 * int(n_conventions) + 1)
    return per if per < multiple else per - per % multiple


def local_segments(threads_total: int, n_conventions: int, world: int, rank: int) -> List[Segment]:
    """The same segment structure on every GPU at 1/world of the threads: each rank
    evaluates every convention pair locally with replicated policy weights."""
    return pairing_segments(KitchenShard(threads_total, world, rank).count, n_conventions)


class TrajectoryExchange:
    """All-gather of compact time-major trajectory shards: ONE collective per update.

    `gather_buffer` sends a `TrajectoryBuffer(store_states=True)`'s flat int32 `wire` buffer -- [records | actions | reward | done],
    the layout of `trajectory.wire_layout`, filled in place by the rollout's own launches -- through a single
    `all_gather_into_tensor` AS IT IS (RCCL over xGMI on the GPU box, gloo in the CPU tests: both take flat 1-D buffers; no
    concatenation pass, no per-field copies) and hands back per-field views of the result, peers in rank order: [world, ...].
    `gather_trajectory` does the same for four loose tensors at the price of one concatenation pass.  Shards must be equal-sized
    (use a kitchen count divisible by the world size); errors of the collective propagate."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, always_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collectives = 0                      # all-gathers issued so far (one per gather_trajectory / gather_buffer)
        # a single rank needs no collective; always_collective issues it anyway (how a 1-GPU box exercises the RCCL call)
        self.always_collective = bool(always_collective) and dist.is_initialized()

    def all_gather(self, local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[...] -> [world, ...], rank r's tensor at index r.  `out`: optional flat destination of world * numel elements
        (a persistent receive buffer: one allocation for all updates)."""
        local = local.contiguous()
        if self.world == 1 and not self.always_collective:
            return local.unsqueeze(0)
        if out is None:
            out = torch.empty((self.world * local.numel(),), dtype=local.dtype, device=local.device)
        elif out.numel() != self.world * local.numel() or out.dtype != local.dtype or not out.is_contiguous():
            raise ValueError("out must be a contiguous buffer of world * numel elements of the shard's dtype")
        dist.all_gather_into_tensor(out.view(-1), local.view(-1), group=self.group)
        self.collectives += 1
        return out.view((self.world,) + tuple(local.shape))

    @staticmethod
    def _as_words(t: torch.Tensor) -> torch.Tensor:
        if t.dtype != torch.int32:
            raise ValueError("compact trajectory fields are int32 (records as int32 words)")
        return t.contiguous().view(-1)

    def gather_buffer(self, traj, out: Optional[torch.Tensor] = None):
        """traj: a TrajectoryBuffer built with store_states=True (every rank with the same T, P, n, layout)
        -> dict of [world, ...] views of ONE gathered buffer: records [world, T+1, n, rec/4], actions / reward [world, T, P, n],
        done [world, T, n].  The shard's `wire` buffer is the send buffer itself."""
        from .trajectory import wire_layout
        if traj.wire is None:
            raise ValueError("gather_buffer needs a TrajectoryBuffer(store_states=True): its fields live in one wire buffer")
        fields, words = wire_layout(traj.T, traj.P, traj.engine.num_kitchens, traj.engine.record_bytes // 4)
        assert words == traj.wire.numel()
        got = self.all_gather(traj.wire, out)     # [world, words]
        res = {}
        for name, (off, shape) in fields.items():
            n = 1
            for d in shape:
                n *= d
            res[name] = got[:, off:off + n].view((got.shape[0],) + shape)
        return res

    def chunked(self, traj, chunk_steps, engine=None, out=None, recv=None, stream=None):
        """The same exchange in chunks of `chunk_steps` steps on a side stream, overlapped with the rollout: see ChunkedGather."""
        return ChunkedGather(self, traj, chunk_steps, engine=engine, out=out, recv=recv, stream=stream)

    def gather_trajectory(self, records, actions, reward, done):
        """Loose fields (one concatenation pass; prefer gather_buffer): records int32 [T, n, rec/4]; actions int32 [T, P, n];
        reward int32 [T, n]; done int32 [T, n] -> dict of [world, ...] views of ONE gathered buffer (peer r's shard at index r)."""
        fields = (("records", records), ("actions", actions), ("reward", reward), ("done", done))
        flat = [self._as_words(t) for _, t in fields]
        sizes = [f.numel() for f in flat]
        pad = (-sum(sizes)) % 4                   # every rank's block starts 16-byte aligned (the encode kernel reads uint4)
        if pad:
            flat.append(torch.zeros(pad, dtype=torch.int32, device=flat[0].device))
        got = self.all_gather(torch.cat(flat))    # [world, words]
        out, off = {}, 0
        for (name, t), sz in zip(fields, sizes):
            out[name] = got[:, off:off + sz].view((got.shape[0],) + tuple(t.shape))
            off += sz
        return out

    def featurize(self, engine, records_all, t, out=None):
        """Observation of every kitchen of every rank at time t from the gathered records:
        int8 [P, world*n, W, H, C], peers' shards in rank order (one encode launch per peer)."""
        world, _, n = records_all.shape[:3]
        if out is None:
            out = torch.empty((engine.num_players, world * n, engine.width, engine.height, engine.channels), dtype=torch.int8,
                              device=records_all.device)
        for r in range(world):
            engine.encode_states_into(records_all[r, t], out, r * n)
        return out

    def featurize_all(self, engine, records_all, out=None):
        """The whole gathered trajectory re-featurised: int8 [T, P, world*n, W, H, C] from records [world, T, n, rec/4],
        ONE launch per peer for all T time steps (oc_encode_trajectory)."""
        world, T, n = records_all.shape[:3]
        if out is None:
            out = torch.empty((T, engine.num_players, world * n, engine.width, engine.height, engine.channels), dtype=torch.int8,
                              device=records_all.device)
        for r in range(world):
            engine.encode_trajectory_into(records_all[r], out, r * n)
        return out


class ChunkedGather:
    """The per-update exchange cut into chunks of `chunk_steps` time steps and OVERLAPPED with the rollout that is still filling the
    wire buffer (consumer: the reference's XP / MP losses read whole buffers once per update, train/XD/xd.py:311-397,
    so nothing needs a peer's step before the update starts -- and nothing stops it from travelling earlier).

    The step launches write the records of step t into slot t + 1 of the wire buffer as they run.  After the last step of chunk c has
    been enqueued the caller says `submit(c)`: an event is recorded on the current (rollout) stream, and on a SIDE stream -- behind
    that event -- the chunk's record slots go through one `all_gather_into_tensor` (RCCL over xGMI; the slots of a chunk are one
    contiguous range of the wire buffer) and, when an engine and an output tensor were given, every peer's block of the chunk is
    re-featurised into the chunk's time slots of the global observation tensor (`oc_encode_trajectory`, one launch per peer).  The
    rollout stream never waits.  `finish()` sends the small fields -- actions | reward | done, the contiguous tail of the wire buffer,
    20 bytes per kitchen-step against the records' 4 S + 16 -- in one more collective and makes the current stream wait for the side
    stream.  Collectives per update: chunks + 1.

    Receive layout: chunk-major -- chunk c's gathered block is `[world, slots_c, n, rec/4]` (peers in rank order), all chunks in one
    flat buffer; `records_all()` assembles the one-shot layout `[world, T + 1, n, rec/4]` of `gather_buffer` for consumers that want
    it (a copy: tests compare it bit for bit with the one-shot gather).  On the CPU (gloo: the tests) everything runs in line."""

    def __init__(self, exchange, traj, chunk_steps, engine=None, out=None, recv=None, stream=None):
        from .trajectory import wire_layout
        if traj.wire is None:
            raise ValueError("ChunkedGather needs a TrajectoryBuffer(store_states=True): its fields live in one wire buffer")
        self.ex, self.traj, self.engine, self.out = exchange, traj, engine, out
        self.T, self.C = int(traj.T), max(1, int(chunk_steps))
        self.n = traj.engine.num_kitchens
        self.recw = traj.engine.record_bytes // 4
        self.fields, words = wire_layout(self.T, traj.P, self.n, self.recw)
        assert words == traj.wire.numel()
        self.n_chunks = (self.T + self.C - 1) // self.C
        world = exchange.world
        # record slots of chunk c: [s0, s1) -- chunk 0 also carries slot 0 (the state the trajectory starts from)
        self.slots = [(0 if c == 0 else c * self.C + 1, min((c + 1) * self.C, self.T) + 1) for c in range(self.n_chunks)]
        slot_words = self.n * self.recw
        need = world * (self.T + 1) * slot_words
        dev = traj.wire.device
        if recv is None:
            recv = torch.empty(need, dtype=torch.int32, device=dev)
        elif recv.numel() < need or recv.dtype != torch.int32 or not recv.is_contiguous():
            raise ValueError(f"recv must be a contiguous int32 buffer of at least {need} words")
        self.recv = recv
        self.chunk_views = []
        for s0, s1 in self.slots:
            off = world * s0 * slot_words
            self.chunk_views.append(recv[off:off + world * (s1 - s0) * slot_words].view(world, s1 - s0, self.n, self.recw))
        self.small = None
        self.cuda = dev.type == "cuda"
        self.stream = (stream if stream is not None else torch.cuda.Stream(dev)) if self.cuda else None
        self.events = [torch.cuda.Event() for _ in range(self.n_chunks)] if self.cuda else None
        self.submitted = 0

    def chunk_of_step(self, t):
        """chunk that the step writing record slot t + 1 belongs to"""
        return min(t // self.C, self.n_chunks - 1)

    def last_step_of(self, c):
        return min((c + 1) * self.C, self.T) - 1

    def _side(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.cuda else contextlib.nullcontext()

    def submit(self, c):
        """Call after the last step of chunk c has been enqueued on the current stream (chunks in order)."""
        if c != self.submitted:
            raise ValueError(f"chunks are submitted in order: expected {self.submitted}, got {c}")
        s0, s1 = self.slots[c]
        off, slot_words = self.fields["records"][0], self.n * self.recw
        send = self.traj.wire[off + s0 * slot_words: off + s1 * slot_words]
        if self.cuda:
            self.events[c].record(torch.cuda.current_stream(send.device))
        with self._side():
            if self.cuda:
                self.stream.wait_event(self.events[c])
            got = self.ex.all_gather(send, out=self.chunk_views[c].view(-1))
            if got.data_ptr() != self.chunk_views[c].data_ptr():          # (one rank without always_collective: all_gather hands the shard back)
                self.chunk_views[c].view(-1).copy_(got.reshape(-1))
            if self.engine is not None and self.out is not None:
                for r in range(self.ex.world):
                    self.engine.encode_trajectory_into(self.chunk_views[c][r], self.out[s0:s1], r * self.n)
        self.submitted = c + 1

    def finish(self):
        """The small fields in one more collective; the current stream then waits for everything the side stream did.
        -> {"records": [per-chunk [world, slots, n, rec/4] views], "actions" / "reward": [world, T, P, n], "done": [world, T, n]}"""
        if self.submitted != self.n_chunks:
            raise ValueError(f"{self.n_chunks - self.submitted} chunk(s) were never submitted")
        a0 = self.fields["actions"][0]
        send = self.traj.wire[a0:]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(send.device))
        with self._side():
            if self.cuda:
                self.stream.wait_event(ev)
            self.small = self.ex.all_gather(send)                           # [world, words of actions | reward | done]
        if self.cuda:
            torch.cuda.current_stream(send.device).wait_stream(self.stream)
        self.submitted = 0                                            # (the next update's rollout may reuse this object and its buffers)
        res = {"records": self.chunk_views}
        for name in ("actions", "reward", "done"):
            off, shape = self.fields[name]
            cnt = 1
            for d in shape:
                cnt *= d
            res[name] = self.small[:, off - a0:off - a0 + cnt].view((self.small.shape[0],) + shape)
        return res

    def records_all(self):
        """[world, T + 1, n, rec/4] in the one-shot layout (a copy)."""
        return torch.cat(self.chunk_views, dim=1)


def segment_scores(episode_returns: torch.Tensor, segments: List[Segment], group: Optional[dist.ProcessGroup] = None):
    """Mean episode return per pairing segment over all ranks: the 2k+1 numbers the reference's score
    bookkeeping needs per episode (`xp_scores`, `best_i`: train/XD/xd_player.py:143-149) -- the only other
    collective of a sharded run, one all-reduce of 2(2k+1) floats.  episode_returns: [n_local] float."""
    sums = torch.stack([episode_returns[s.start:s.stop].sum() for s in segments])
    counts = torch.tensor([float(s.stop - s.start) for s in segments], device=episode_returns.device, dtype=sums.dtype)
    packed = torch.cat([sums, counts])
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, group=group)
    g = len(segments)
    return packed[:g] / packed[g:]


def best_partner(scores: torch.Tensor, n_conventions: int, xp_weight: float = 1.0) -> int:
    """argmax_i of xp_weight * (mean XP0_i + mean XP1_i) (train/XD/xd_player.py:144-149)."""
    k = n_conventions
    both = xp_weight * (scores[1:1 + k] + scores[1 + k:1 + 2 * k])
    return int(both.argmax())


def segment_policy_indices(segments, side: str) -> list:
    """Policy index per pairing segment for 'ego' or 'partner' (see sharding.pairing_segments)."""
    return [s.ego_policy if side == "ego" else s.partner_policy for s in segments]


def segment_critics(segments, side: str) -> list:
    """Which critic VALUES each pairing segment on a side while the reference collects (pinned to its own runner by
    tests/golden/pairing_cases.npz): the convention being trained always brings its own current critic ("new": `policy.get_actions`,
    train/XD/xd_player.py:94,190-197 and :76-77), a frozen convention i on the ego side is wrapped with `xp_critic0[i]`
    (xd_player.py:92), on the partner side with `xp_critic1[i]` (xd_player.py:74).  -> ["new" | ("xp0", i) | ("xp1", i)] per segment."""
    out = []
    for s in segments:
        pol = s.ego_policy if side == "ego" else s.partner_policy
        out.append("new" if pol == 0 else (("xp0" if side == "ego" else "xp1"), pol - 1))
    return out
