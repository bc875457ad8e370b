"""Mixed-play data collection on the device, generalised from the reference's L-1 environments
to any number of kitchens.

Reference (train/XD/xd_player.py:232-356 `collect_mp_episode` / `next_mp_step`, train/partner_agents.py:153-244
`MixedAgent`, train/MAPPO/utils/shared_buffer.py:150-220 `diaginsert` / `partinsert`): `env_length - 1` environments
run two episodes of L = episode_length steps.  On each side every environment is played, step by step, by a coin
flip between the policy being trained (0) and the frozen partner convention (1), except that environment e is
forced onto policy 0 from step L-1-e of the first episode and from step e+1 of the second: a staircase of
switch times.  Only the forced stretches are kept, and `diaginsert` / `partinsert` shift them so that row e of the
`[L, L-1]` buffer is [end of episode 1 | rest of episode 2].  All of that is index arithmetic; here it is:

* `MixedPlaySchedule`: kitchen k takes the role of environment `k // ceil(N / (L-1))` (contiguous stretches); `choose(step, uniforms)` is the
  reference's `mix_mask` for a whole `[N]` vector on the device, `gather_index()` the placement map of the two
  inserts as one row index, so the mixed-play buffer is ONE gather out of a time-aligned trajectory.
* `MixedPlayRollout`: the 2L-step collection itself -- every step's rows are sorted by who plays them (one stable partition of
  the whole `[2L, P, N]` choice tensor on the device, `oc_partition_rows`), and per step one fused policy launch in which each of
  the two candidate networks (actor and critic) evaluates exactly the rows the coin gave it, gathered through the index lists
  (`OC_POLICY_TILE_GATHER`: the reference's `mix_mask == i` indexing, no row is evaluated twice), and one env launch writing the
  next trajectory slot.
Pinned by tests/golden/mixed_play_cases.npz, recorded from the reference's own classes.
"""
import torch

from .engine import OvercookedEngine
from .fused_policy import TILE_ACT, GatherPlan
from .rollout_ops import gather_rows
from .trajectory import TrajectoryBuffer


class MixedPlaySchedule:
    def __init__(self, length: int, n_kitchens: int, device="cpu"):
        if length < 2:
            raise ValueError("episode length must be at least 2")
        self.L, self.N, self.device = int(length), int(n_kitchens), torch.device(device)
        # the reference's environment index, in contiguous stretches of kitchens: the kitchens already forced onto the
        # trained policy then form one range, whose 32-row blocks the partner networks' tiles skip outright
        reps = -(-self.N // (self.L - 1))
        self.lane = torch.arange(self.N, device=self.device) // reps

    def forced_main(self, step: int) -> torch.Tensor:
        """bool [N]: kitchens whose `mix_mask` entry the reference zeroes at wall step `step` (0 .. 2L-1)."""
        L, e = self.L, self.lane
        if step >= L:
            return e < (step - L)                     # mix_mask[:step - L] = 0
        if step > 0:
            return e >= (L - 1 - step)                # mix_mask[-step:] = 0
        return torch.zeros_like(e, dtype=torch.bool)

    def choose(self, step: int, uniforms: torch.Tensor) -> torch.Tensor:
        """uniforms f32 [..., N] -> int32 [..., N]: 1 where the partner convention plays, 0 where the trained policy does."""
        return ((uniforms < 0.5) & ~self.forced_main(step)).to(torch.int32)

    def choose_all(self, uniforms: torch.Tensor) -> torch.Tensor:
        """uniforms f32 [2L, ..., N] -> int32 [2L, ..., N]: `choose` for every wall step at once."""
        L, e = self.L, self.lane
        step = torch.arange(2 * L, device=self.device).view(2 * L, *([1] * (uniforms.dim() - 1)))
        forced = torch.where(step >= L, e < (step - L), (step > 0) & (e >= (L - 1 - step)))
        return ((uniforms < 0.5) & ~forced).to(torch.int32)

    def source_step(self) -> torch.Tensor:
        """int64 [L, N]: wall step whose turn data the reference stores at buffer time t of kitchen k."""
        L = self.L
        t = torch.arange(L, device=self.device).unsqueeze(1)
        e = self.lane.unsqueeze(0)
        return torch.where(t <= e, (L - 1 - e) + t, L + t)                           # diaginsert | partinsert

    def gather_index(self, players: int = 2) -> torch.Tensor:
        """int64 [L * P * N]: row of a time-aligned `[2L(+1), P, N]` store for every entry of the `[L, P, N]` buffer."""
        src = self.source_step().unsqueeze(1)                                        # [L, 1, N]
        p = torch.arange(players, device=self.device).view(1, players, 1)
        k = torch.arange(self.N, device=self.device).view(1, 1, self.N)
        return ((src * players + p) * self.N + k).reshape(-1).contiguous()


class MixedPlayRollout:
    """main / partner: (actor net index, critic net index) in `nets`; the reference's `policies = [self.policy,
    rMAPPOWrapper(agent_set[best_i], policy.mp_critic)]` (xd_player.py:235)."""

    def __init__(self, engine: OvercookedEngine, nets, length: int, main, partner, ego_id: int = 0, seed: int = 0,
                 deterministic: bool = False, generator=None):
        if int(engine.params["horizon"]) != int(length):
            raise ValueError(f"the engine's horizon ({engine.params['horizon']}) must equal the episode length ({length}): "
                             "both mixed-play episodes end for every kitchen at once")
        self.engine, self.nets, self.L = engine, nets, int(length)
        n = engine.num_kitchens
        self.schedule = MixedPlaySchedule(length, n, engine.device)
        tiles = []
        for side in range(2):
            for choice, (actor, critic) in enumerate((main, partner)):
                tiles.append((side, choice, actor, TILE_ACT))
                tiles.append((side, choice, critic, 0))
        self.plan = GatherPlan(nets, 2 * n, n, tiles)
        self.ego_id, self.seed, self.deterministic, self.generator = ego_id, seed, deterministic, generator
        # running draw counter: the kernel's draw uses hash(seed, counter + t, row, 0), so every collection
        # must start where the previous one stopped or it would replay the same noise (as FusedPairedRollout.steps_done does)
        self.draws_done = 0
        T, dev = 2 * self.L, engine.device
        self.traj = TrajectoryBuffer(engine, T)
        self.logp = torch.zeros((T, 2, n), dtype=torch.float32, device=dev)
        self.values = torch.zeros((T, 2, n), dtype=torch.float32, device=dev)
        self.choice = torch.zeros((T, 2, n), dtype=torch.int32, device=dev)
        self.index = torch.zeros((T, 2, n), dtype=torch.int32, device=dev)          # rows of every (step, side) sorted by choice
        self.count = torch.zeros((T, 2, 2), dtype=torch.int32, device=dev)

    def collect(self, uniforms=None):
        """Run the two episodes.  uniforms: optional f32 [2L, 2, N] (index 0 = the ego side's draw, 1 = the partner's,
        the order the reference draws them in); default: `torch.rand` on the device."""
        import ctypes
        from . import native
        from .fused_policy import _ptr, _shape, _stream
        e, n, T, nets = self.engine, self.engine.num_kitchens, 2 * self.L, self.nets
        e.reset_all()
        self.traj.start()
        if uniforms is None:
            uniforms = torch.rand((T, 2, n), device=e.device, generator=self.generator)
        ch = self.schedule.choose_all(uniforms)                                         # [2L, 2, N]: ego draw, partner draw
        self.choice[:, self.ego_id], self.choice[:, 1 - self.ego_id] = ch[:, 0], ch[:, 1]
        # all 2L steps (the partition of every step's rows, then per step a gathered policy launch + the env launch) enqueued by one call
        native.check(native.lib().oc_policy_rollout_mixed(
            e._handle, ctypes.byref(_shape(nets.W, nets.H, 6, nets.digits)), _ptr(nets.image), nets.n_nets, _ptr(self.plan.tiles),
            self.plan.n_tiles, T, _ptr(self.traj.obs), _ptr(self.traj.actions), _ptr(self.logp), _ptr(self.values),
            _ptr(self.traj.reward), _ptr(self.traj.done), _ptr(self.traj.records), 0 if self.deterministic else 1, int(self.seed) & 0xFFFFFFFF,
            self.draws_done & 0xFFFFFFFF, _ptr(self.choice), _ptr(self.index), _ptr(self.count), _stream(e.device)))
        self.draws_done += T
        self.traj.t = T
        return self

    def buffer(self):
        """The mixed-play buffer in the reference's time order, engine layout: obs int8 [L, P, N, W, H, C]; actions int32,
        action_log_probs / value_preds f32, rewards int32 [L, P, N]; masks f32 [L, P, N] (0 where that step ended the
        episode, what `turn_mp_masks` holds when the reference inserts it)."""
        L, n, tr = self.L, self.engine.num_kitchens, self.traj
        idx = self.schedule.gather_index(2)
        obs = gather_rows(tr.obs.view(-1, *tr.obs.shape[3:]), idx).view(L, 2, n, *tr.obs.shape[3:])
        take = lambda t: t.reshape(-1).index_select(0, idx).view(L, 2, n)
        done = tr.done.unsqueeze(1).expand(-1, 2, -1)
        return dict(obs=obs, actions=take(tr.actions), action_log_probs=take(self.logp), value_preds=take(self.values),
                    rewards=take(tr.reward), masks=take((1 - done).to(torch.float32).contiguous()),
                    choice=take(self.choice), source_step=self.schedule.source_step())
