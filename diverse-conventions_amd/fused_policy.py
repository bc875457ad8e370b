"""Fused policy inference: front-end of `oc_policy_forward`.

The reference walks the pairing segments and calls `policy.get_actions` -- an R_Actor and an R_Critic forward of
several launches each -- once per segment and side (train/XD/xd_player.py:190-207, train/partner_agents.py:97-111).
Here every network of a rollout step (all conventions' actors and critics, both sides) is one tile list for ONE
launch of a hand-written MFMA kernel that reads the engine's int8 observations and writes the int32 actions the
step kernel reads.  `PackedNetworks` holds the networks in the kernel's weight image (packed on the device,
stream-ordered, so it can follow an optimiser step); `PolicyPlan` is the tile list for a fixed segment -> network
map.  There is no fallback: without the HIP library these classes raise.
"""
import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import native

ACTOR_HEAD = "act.action_out.linear"        # reference utils/act.py + distributions.py:60-80
CRITIC_HEAD = "v_out"                       # reference r_actor_critic.py:165
TILE_ACT = 1                                # OC_POLICY_TILE_ACT
TILE_GATHER = 16                            # OC_POLICY_TILE_GATHER: the tile is (list, class, net, flags) over an index list


def tile_select(v):
    """OC_POLICY_TILE_SELECT(v): the tile writes only rows whose `row_select` entry equals v."""
    return ((int(v) + 1) & 0xFF) << 8

_KEYS = ("base.cnn.cnn.0.weight", "base.cnn.cnn.0.bias", "base.cnn.cnn.3.weight", "base.cnn.cnn.3.bias",
         "base.cnn.cnn.5.weight", "base.cnn.cnn.5.bias")


def _shape(width, height, n_out, digits, hidden=64):
    return native.OcPolicyShape(int(width), int(height), int(hidden), int(n_out), int(digits))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    from .engine import _raw_stream                     # (the raw handle: no torch.cuda.Stream object per call)
    d = torch.device(device)
    return _raw_stream(d.index if d.index is not None else torch.cuda.current_device())


DEFAULT_DIGITS = 3          # base-256 conv weight digits: 24 bits, the mantissa of the f32 weights (csrc/policy_pack.h)


def packed_bytes(width, height, digits=DEFAULT_DIGITS):
    n = native.lib().oc_policy_packed_bytes(ctypes.byref(_shape(width, height, 1, digits)))
    if n < 0:
        native.check(int(n))
    return int(n)


def pack_host(state_dict, obs_shape, head=ACTOR_HEAD, digits=DEFAULT_DIGITS) -> np.ndarray:
    """Weight image of one reference-format network as a uint8 array (host packer; tests, inspection)."""
    W, H = int(obs_shape[0]), int(obs_shape[1])
    arrs = [np.ascontiguousarray(np.asarray(state_dict[k], dtype=np.float32)) for k in _KEYS]
    arrs += [np.ascontiguousarray(np.asarray(state_dict[head + s], dtype=np.float32)) for s in (".weight", ".bias")]
    hidden = arrs[4].shape[0]
    out = np.zeros(packed_bytes(W, H, digits), dtype=np.uint8)
    native.check(native.lib().oc_policy_pack_host(ctypes.byref(_shape(W, H, arrs[6].shape[0], digits, hidden)),
                                                  *[a.ctypes.data_as(ctypes.c_void_p) for a in arrs],
                                                  out.ctypes.data_as(ctypes.c_void_p)))
    return out


ACTIVATION_LIMIT = 65504.0 / 16.0   # = 4094: where the kernel's hidden activations saturate (PackedNetworks)


class PackedNetworks:
    """`n_nets` networks of one observation shape in the kernel's weight image, resident in HBM.

    `pack(i, state_dict, head)` converts reference-format tensors (host or device) on the device; call it again
    after every optimiser step for the networks that changed.

    Range.  The dense layers run on f16 two-piece operands: every hidden activation (conv, fc1, fc2 outputs after the
    ReLU) travels at 2^4 x its value and is clamped at 65 504, i.e. **activations above `ACTIVATION_LIMIT` = 4094 saturate** where the
    reference's f32 forward does not, and a layer whose weights span more than ~2^14 in magnitude loses the low bits of its
    smallest weights (below 2^-24 of the layer's largest product).  Networks of the reference's training stay orders of magnitude
    inside both (activations of O(1-100)); `activation_range(i, obs)` evaluates network i in f32 with PyTorch on the rows given and
    reports the largest activation per layer; `from_state_dicts` and `check_finite()` (or `pack(..., check_finite=True)`) refuse NaN / Inf
    weights -- the device-side scale search skips non-finite weights, so they would otherwise pack into garbage silently."""

    def __init__(self, obs_shape, n_nets, device, digits=DEFAULT_DIGITS):
        self.W, self.H = int(obs_shape[0]), int(obs_shape[1])
        if len(obs_shape) > 2 and int(obs_shape[2]) != 20:
            raise ValueError("the fused kernel reads the engine's 20-channel observations")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("PackedNetworks lives on the GPU (the fused policy kernel has no CPU path)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.digits, self.n_nets = int(digits), int(n_nets)
        self.bytes_per_net = packed_bytes(self.W, self.H, self.digits)
        self.image = torch.zeros((self.n_nets, self.bytes_per_net), dtype=torch.uint8, device=self.device)
        self.n_out = [0] * self.n_nets
        self._keep = {}

    def pack(self, index, state_dict, head=ACTOR_HEAD, check_finite=False):
        """state_dict: an R_Actor / R_Critic of the reference's default configuration -- CNNBase with ReLU (`--use_ReLU`
        left at its default; the activation has no parameters, so a Tanh network cannot be told apart here) and no
        recurrent layer (`--use_recurrent_policy` off, as in every training script of the reference).
        check_finite: raise ValueError on NaN / Inf parameters of THIS network before packing (reductions and a host read: ~0.2 ms
        against the 40-60 us of the pack itself, so off by default for re-packs behind an optimiser step; `from_state_dicts` checks
        everything it packed with one read, `check_finite()` does the same for the networks packed since)."""
        if any(k.startswith("rnn.") for k in state_dict):
            raise NotImplementedError("recurrent policies (rnn.* parameters) are not evaluated by the fused kernel; "
                                      "the reference's training scripts do not enable them")
        t = lambda k: torch.as_tensor(state_dict[k]).detach().to(device=self.device, dtype=torch.float32).contiguous()
        src = [t(k) for k in _KEYS] + [t(head + ".weight"), t(head + ".bias")]
        hidden, n_out = src[4].shape[0], src[6].shape[0]
        P = (self.W - 2) * (self.H - 2)
        if tuple(src[0].shape) != (hidden // 2, 20, 3, 3) or tuple(src[2].shape) != (hidden, hidden // 2 * P):
            raise ValueError(f"network {index} does not have the reference CNNBase shape for a {self.W}x{self.H} grid")
        if check_finite:
            self._raise_if_nonfinite({index: (src, head)})
        self._keep[index] = src                      # the pack kernel reads them asynchronously
        self._heads = getattr(self, "_heads", {})
        self._heads[index] = head
        native.check(native.lib().oc_policy_pack(ctypes.byref(_shape(self.W, self.H, n_out, self.digits, hidden)),
                                                 *[_ptr(x) for x in src], _ptr(self.image[index]), _stream(self.device)))
        self.n_out[index] = n_out
        return index

    @staticmethod
    def _raise_if_nonfinite(nets):
        """nets: {index: (tensors, head)} -> ValueError naming the first network and parameters that hold NaN / Inf.  One fused norm
        over all tensors and ONE host read when everything is finite."""
        flat = [x for src, _ in nets.values() for x in src]
        try:
            norms = torch.stack(torch._foreach_norm(flat))
        except (AttributeError, RuntimeError):
            norms = torch.stack([x.abs().max() if x.numel() else x.new_zeros(()) for x in flat])
        if bool(norms.isfinite().all()):
            return
        for index, (src, head) in nets.items():
            bad = [k for k, x in zip(_KEYS + (head + ".weight", head + ".bias"), src) if not bool(x.isfinite().all())]
            if bad:
                raise ValueError(f"network {index}: non-finite values in {', '.join(bad)}")

    def check_finite(self, indices=None):
        """Raise ValueError if any parameter handed to `pack` (for `indices`, default: every packed network) holds NaN / Inf: one
        device reduction and one host read for all of them.  The device-side scale search skips non-finite weights, so a network
        that has them packs into garbage silently -- call this after re-packing networks from an optimiser you do not trust."""
        idx = list(self._keep) if indices is None else list(indices)
        self._raise_if_nonfinite({i: (self._keep[i], self._heads[i]) for i in idx})

    def activation_range(self, index, obs):
        """Largest hidden activation of network `index` on the rows `obs` (int8 / float [n, W, H, 20], any device), evaluated in f32
        by PyTorch from the tensors `pack` was given (train/MAPPO/utils/cnn.py:26-38) -> {"conv", "fc1", "fc2": max after the ReLU,
        "out": max |head output|, "limit": ACTIVATION_LIMIT, "saturates": any hidden layer above the limit}.  An audit to run on a
        sample of real observations when networks come from somewhere else than the reference's training (hand-set weights,
        other initialisations): above the limit the fused kernel's outputs differ from the reference's."""
        cw, cb, w1, b1, w2, b2, wh, bh = self._keep[index]
        x = torch.as_tensor(obs).to(device=self.device, dtype=torch.float32).reshape(-1, self.W, self.H, 20).permute(0, 3, 1, 2)
        c = torch.relu(torch.nn.functional.conv2d(x, cw, cb))                 # Conv2d on [B, C, W, H], flatten (o, x', y')
        f1 = torch.relu(torch.nn.functional.linear(c.flatten(1), w1, b1))
        f2 = torch.relu(torch.nn.functional.linear(f1, w2, b2))
        out = torch.nn.functional.linear(f2, wh, bh)
        r = {"conv": float(c.max()), "fc1": float(f1.max()), "fc2": float(f2.max()), "out": float(out.abs().max()), "limit": ACTIVATION_LIMIT}
        r["saturates"] = max(r["conv"], r["fc1"], r["fc2"]) > ACTIVATION_LIMIT
        return r

    @classmethod
    def from_state_dicts(cls, obs_shape, device, actors: Sequence[dict] = (), critics: Sequence[dict] = (), digits=DEFAULT_DIGITS):
        """Actors get indices 0..len(actors)-1, critics follow."""
        nets = cls(obs_shape, len(actors) + len(critics), device, digits)
        for i, sd in enumerate(actors):
            nets.pack(i, sd, ACTOR_HEAD)
        for i, sd in enumerate(critics):
            nets.pack(len(actors) + i, sd, CRITIC_HEAD)
        nets.check_finite()                          # (one read for all of them)
        return nets


def merge_segments(segments):
    """Row ranges of one network with the same flags that touch become one segment: the pairing layout puts the main policy on a
    run of consecutive segments of each side (train/XD/xd_player.py:76-77,94), and a merged run has no ragged 32-row block and no
    weight reload at every internal boundary (config 5: 130-kitchen segments are four full blocks and a fifth with two rows).
    Segments keep the order in which their (network, flags) first appears."""
    groups = {}
    for row0, rows, net, flags in (tuple(int(v) for v in s) for s in segments):
        groups.setdefault((net, flags), []).append((row0, rows))
    out = []
    for (net, flags), spans in groups.items():
        merged = []
        for row0, rows in sorted(spans):
            if merged and merged[-1][0] + merged[-1][1] == row0:
                merged[-1] = (merged[-1][0], merged[-1][1] + rows)
            else:
                merged.append((row0, rows))
        out += [(row0, rows, net, flags) for row0, rows in merged]
    return out


def plan_tiles(segments, rows_per_tile):
    """segments: iterable of (row0, rows, net, flags) -> int32 [n_tiles, 4] host tensor (oc_policy_plan_tiles)."""
    seg = np.asarray(list(segments), dtype=np.int64).reshape(-1, 4)
    row0, rows = np.ascontiguousarray(seg[:, 0]), np.ascontiguousarray(seg[:, 1])
    net, flags = np.ascontiguousarray(seg[:, 2].astype(np.int32)), np.ascontiguousarray(seg[:, 3].astype(np.int32))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    L = native.lib()
    n = L.oc_policy_plan_tiles(p(row0), p(rows), p(net), p(flags), len(seg), int(rows_per_tile), None, 0)
    if n < 0:
        native.check(int(n))
    tiles = np.zeros((max(int(n), 1), 4), dtype=np.int32)
    n2 = L.oc_policy_plan_tiles(p(row0), p(rows), p(net), p(flags), len(seg), int(rows_per_tile), p(tiles), int(n))
    if n2 < 0:
        native.check(int(n2))
    return torch.from_numpy(tiles[:int(n)])


class PolicyPlan:
    """Tile list for a fixed (row range -> network) map over `n_rows` observation rows.

    segments: (row0, rows, net, flags); flags TILE_ACT = actor epilogue (actions + log-probs), 0 = raw head outputs
    (critic values).  Rows are flat indices into the engine's `[P, N]` leading axes: row = player * N + kitchen."""

    def __init__(self, nets: PackedNetworks, n_rows: int, segments, rows_per_tile: Optional[int] = None):
        self.nets, self.n_rows = nets, int(n_rows)
        segments = [tuple(int(v) for v in s) for s in segments]
        for row0, rows, net, _ in segments:
            if row0 < 0 or rows < 0 or row0 + rows > self.n_rows or not 0 <= net < nets.n_nets:
                raise ValueError(f"segment ({row0}, {rows}, net {net}) outside {self.n_rows} rows / {nets.n_nets} networks")
        segments = merge_segments(segments)
        if rows_per_tile is None:
            # balanced plan: one tile per CU, pieces of one segment (= one network) each
            rows_per_tile = -torch.cuda.get_device_properties(nets.device).multi_processor_count
        self.rows_per_tile = int(rows_per_tile)
        self.tiles = plan_tiles(segments, self.rows_per_tile).to(nets.device)
        self.n_tiles = int(self.tiles.shape[0])
        self.counter = 0

    def forward(self, obs, raw=None, actions=None, action_logp=None, logp=None, deterministic=False, seed=0, counter=None,
                device_counter=None, row_select=None):
        """obs: contiguous int8 tensor whose leading axes flatten to n_rows rows of W*H*20 bytes."""
        nets = self.nets
        rb = nets.W * nets.H * 20
        if obs.dtype != torch.int8 or not obs.is_contiguous() or obs.device != nets.device or obs.numel() != self.n_rows * rb:
            raise ValueError(f"obs must be a contiguous int8 tensor of {self.n_rows} rows x {rb} bytes on {nets.device}")
        for name, t, dt in (("raw", raw, torch.float32), ("actions", actions, torch.int32),
                            ("action_logp", action_logp, torch.float32), ("logp", logp, torch.float32)):
            if t is not None and (t.dtype != dt or not t.is_contiguous() or t.device != nets.device):
                raise ValueError(f"{name} must be a contiguous {dt} tensor on {nets.device}")
        if device_counter is not None:
            if device_counter.dtype != torch.int32 or device_counter.device != nets.device or device_counter.numel() != 1:
                raise ValueError("device_counter must be a one-element int32 tensor on the networks' device")
            counter = 0 if counter is None else counter
        if row_select is not None and (row_select.dtype != torch.int32 or not row_select.is_contiguous() or row_select.device != nets.device
                                       or row_select.numel() != self.n_rows):
            raise ValueError(f"row_select must be a contiguous int32 tensor of {self.n_rows} entries on {nets.device}")
        if counter is None:
            counter, self.counter = self.counter, self.counter + 1
        shape = _shape(nets.W, nets.H, 6, nets.digits)
        native.check(native.lib().oc_policy_forward(ctypes.byref(shape), _ptr(nets.image), nets.n_nets, _ptr(obs), self.n_rows,
                                                    _ptr(self.tiles), self.n_tiles, _ptr(raw), _ptr(actions), _ptr(action_logp),
                                                    _ptr(logp), 0 if deterministic else 1, int(seed) & 0xFFFFFFFF,
                                                    int(counter) & 0xFFFFFFFF, _ptr(device_counter), _ptr(row_select), _stream(nets.device)))


def tile_slot(i):
    """OC_POLICY_TILE_SLOT(i): the tile belongs to slot (grid) i of a multi-grid launch."""
    return (int(i) & 7) << 16


class MultiPolicyPlan:
    """ONE launch over several observation tensors of different grids (`oc_policy_forward_multi`): the layouts of an engine
    group.  Five per-layout launches cannot share a CU (each fills its registers and LDS), so at 8192 kitchens per layout they
    leave half the SIMDs without a 32-row block; one tile table balanced by block cost does not.

    nets[i]: PackedNetworks of grid i; n_rows[i]; segments[i]: (row0, rows, net, flags) inside grid i."""

    def __init__(self, nets: Sequence[PackedNetworks], n_rows: Sequence[int], segments, workgroups: Optional[int] = None,
                 waves_per_workgroup: int = 0):
        """waves_per_workgroup: 0 lets the planner choose (wave tiles for tables of short segments), 1 forces one-wavefront workgroups
        ("wave tiles"), 4 four-wavefront workgroups."""
        self.nets, self.n_rows = list(nets), [int(n) for n in n_rows]
        if not 1 <= len(self.nets) <= 8 or len(self.n_rows) != len(self.nets) or len(segments) != len(self.nets):
            raise ValueError("1..8 grids, with one row count and one segment list each")
        dev = self.nets[0].device
        if any(nt.device != dev or nt.digits != self.nets[0].digits for nt in self.nets):
            raise ValueError("all grids' networks must live on one device and use the same number of conv digits")
        rows = []
        for i, (nt, n, segs) in enumerate(zip(self.nets, self.n_rows, segments)):
            for row0, cnt, net, flags in merge_segments(segs):
                if row0 < 0 or cnt < 0 or row0 + cnt > n or not 0 <= net < nt.n_nets:
                    raise ValueError(f"grid {i}: segment ({row0}, {cnt}, net {net}) outside {n} rows / {nt.n_nets} networks")
                rows.append((int(row0), int(cnt), int(net), int(flags), i))
        seg = np.asarray(rows, dtype=np.int64).reshape(-1, 5)
        cols = [np.ascontiguousarray(seg[:, 0]), np.ascontiguousarray(seg[:, 1])] + [np.ascontiguousarray(seg[:, j].astype(np.int32)) for j in (2, 3, 4)]
        # relative cost of a 32-row block: conv positions + the position-independent part (measured launches)
        cost = np.asarray([(nt.W - 2) * (nt.H - 2) + 2.5 for nt in self.nets], dtype=np.float32)
        if workgroups is None:
            workgroups = torch.cuda.get_device_properties(dev).multi_processor_count
        self.workgroups = int(workgroups)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        L = native.lib()
        args = [p(c) for c in cols] + [p(cost), len(self.nets), len(seg), self.workgroups]
        per, waves = ctypes.c_int32(0), ctypes.c_int32(int(waves_per_workgroup))
        n = L.oc_policy_plan_tiles_multi(*args, None, 0, ctypes.byref(per), ctypes.byref(waves))
        if n < 0:
            native.check(int(n))
        tiles = np.zeros((max(int(n), 1), 4), dtype=np.int32)
        waves = ctypes.c_int32(int(waves_per_workgroup))
        n2 = L.oc_policy_plan_tiles_multi(*args, p(tiles), int(n), ctypes.byref(per), ctypes.byref(waves))
        if n2 < 0:
            native.check(int(n2))
        self.tiles = torch.from_numpy(tiles[:int(n)]).to(dev)
        self.n_tiles, self.tiles_per_workgroup, self.waves_per_workgroup = int(n), int(per.value), int(waves.value)
        self.counter = 0
        self._shapes = [_shape(nt.W, nt.H, 6, nt.digits) for nt in self.nets]

    def forward(self, obs, raw=None, actions=None, action_logp=None, logp=None, row_select=None, deterministic=False, seeds=None, counter=None,
                device_counter=None):
        """Per-grid lists: obs[i] contiguous int8 with n_rows[i] rows; outputs as PolicyPlan.forward takes them (None or lists)."""
        k = len(self.nets)
        none = [None] * k
        raw, actions, action_logp, logp, row_select = (x if x is not None else none for x in (raw, actions, action_logp, logp, row_select))
        seeds = [0] * k if seeds is None else list(seeds)
        slots = (native.OcPolicyMultiSlot * k)()
        for i, nt in enumerate(self.nets):
            rb = nt.W * nt.H * 20
            if obs[i].dtype != torch.int8 or not obs[i].is_contiguous() or obs[i].device != nt.device or obs[i].numel() != self.n_rows[i] * rb:
                raise ValueError(f"obs[{i}] must be a contiguous int8 tensor of {self.n_rows[i]} rows x {rb} bytes on {nt.device}")
            for name, t, dt in (("raw", raw[i], torch.float32), ("actions", actions[i], torch.int32), ("action_logp", action_logp[i], torch.float32),
                                ("logp", logp[i], torch.float32), ("row_select", row_select[i], torch.int32)):
                if t is not None and (t.dtype != dt or not t.is_contiguous() or t.device != nt.device):
                    raise ValueError(f"{name}[{i}] must be a contiguous {dt} tensor on {nt.device}")
            s = slots[i]
            s.shape = ctypes.pointer(self._shapes[i])
            s.d_packed, s.n_nets, s.d_obs, s.n_rows = _ptr(nt.image), nt.n_nets, _ptr(obs[i]), self.n_rows[i]
            s.d_raw, s.d_actions, s.d_action_logp, s.d_logp, s.d_row_select = (_ptr(raw[i]), _ptr(actions[i]), _ptr(action_logp[i]), _ptr(logp[i]),
                                                                              _ptr(row_select[i]))
            s.seed = int(seeds[i]) & 0xFFFFFFFF
        if counter is None:
            counter, self.counter = self.counter, self.counter + 1
        dev = self.nets[0].device
        native.check(native.lib().oc_policy_forward_multi(slots, k, _ptr(self.tiles), self.n_tiles, self.tiles_per_workgroup, self.waves_per_workgroup,
                                                          0 if deterministic else 1, int(counter) & 0xFFFFFFFF, _ptr(device_counter), _stream(dev)))


def partition_rows(select, id_period=1, index=None, count=None):
    """oc_partition_rows: select int32 [..., len] of 0 / non-0 choices (leading axes = lists) -> (index int32 same shape: per list
    the ids of the entries with choice 0 in ascending order, then the others'; count int32 [..., 2]).  The id of entry k of list
    l is (l % id_period) * len + k."""
    if select.dtype != torch.int32 or not select.is_contiguous() or not select.is_cuda:
        raise ValueError("select must be a contiguous int32 tensor on the GPU")
    length = select.shape[-1]
    lists = select.numel() // length
    index = torch.empty_like(select) if index is None else index
    count = torch.empty(tuple(select.shape[:-1]) + (2,), dtype=torch.int32, device=select.device) if count is None else count
    native.check(native.lib().oc_partition_rows(_ptr(select), lists, length, int(id_period), _ptr(index), _ptr(count), _stream(select.device)))
    return index, count


class GatherPlan:
    """Tile table of a gathered launch (`oc_policy_forward_gather`): `lists` row lists of `length` ids each -- e.g. one per side of
    the engine's `[P, N]` rows -- every one split in two classes by `partition_rows`; tiles = (list, class, net, flags): network
    `net` evaluates the rows of `class` in `list`, and nothing else.  The kernel splits the table evenly over the CUs itself (the
    class counts live on the device), so the table is the same for every step."""

    def __init__(self, nets: PackedNetworks, n_rows: int, length: int, tiles):
        self.nets, self.n_rows, self.length = nets, int(n_rows), int(length)
        rows = []
        for lst, cls, net, flags in tiles:
            if not 0 <= int(net) < nets.n_nets or int(cls) not in (0, 1) or int(lst) < 0:
                raise ValueError(f"gather tile ({lst}, {cls}, net {net}) is not (list >= 0, class 0/1, one of {nets.n_nets} networks)")
            rows.append((int(lst), int(cls), int(net), (int(flags) & 0xFF) | TILE_GATHER))
        self.tiles = torch.tensor(rows, dtype=torch.int32).reshape(-1, 4).to(nets.device)
        self.n_tiles = len(rows)
        self.counter = 0

    def forward(self, obs, index, count, raw=None, actions=None, action_logp=None, logp=None, deterministic=False, seed=0, counter=None,
                device_counter=None):
        nets = self.nets
        rb = nets.W * nets.H * 20
        if obs.dtype != torch.int8 or not obs.is_contiguous() or obs.device != nets.device or obs.numel() != self.n_rows * rb:
            raise ValueError(f"obs must be a contiguous int8 tensor of {self.n_rows} rows x {rb} bytes on {nets.device}")
        for name, t in (("index", index), ("count", count)):
            if t.dtype != torch.int32 or not t.is_contiguous() or t.device != nets.device:
                raise ValueError(f"{name} must be a contiguous int32 tensor on {nets.device}")
        if index.numel() % self.length or count.numel() != 2 * (index.numel() // self.length):
            raise ValueError("index holds whole lists of `length` ids, count two entries per list")
        if int(self.tiles[:, 0].max()) >= index.numel() // self.length:
            raise ValueError("a tile names a list the index array does not hold")
        for name, t, dt in (("raw", raw, torch.float32), ("actions", actions, torch.int32),
                            ("action_logp", action_logp, torch.float32), ("logp", logp, torch.float32)):
            if t is not None and (t.dtype != dt or not t.is_contiguous() or t.device != nets.device):
                raise ValueError(f"{name} must be a contiguous {dt} tensor on {nets.device}")
        if counter is None:
            counter, self.counter = self.counter, self.counter + 1
        native.check(native.lib().oc_policy_forward_gather(
            ctypes.byref(_shape(nets.W, nets.H, 6, nets.digits)), _ptr(nets.image), nets.n_nets, _ptr(obs), self.n_rows, _ptr(self.tiles),
            self.n_tiles, _ptr(index), _ptr(count), self.length, _ptr(raw), _ptr(actions), _ptr(action_logp), _ptr(logp),
            0 if deterministic else 1, int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF, _ptr(device_counter), _stream(nets.device)))


def draw_uniform(seed, counter, row, action=0):
    """The counter-hash uniform of (seed, counter, row, action).  The kernel draws a row's action by inverse CDF from the uniform of
    action 0 (the first action whose cumulative probability exceeds it): lets a test re-derive every sampled action."""
    return float(native.lib().oc_policy_uniform(int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF, int(row), int(action)))


class FusedVectorAgent:
    """A partner (or ego) agent for the reference-style loop -- `env.add_partner_agent(agent)`, `agent.get_action(vobs)`
    (pantheonrl_extension/vectoragent.py; the reference's CentralizedMultiAgent, train/partner_agents.py:78-150, is the
    multi-convention case) -- whose policies run on the fused MFMA kernel: `segments` = (first kitchen, kitchens,
    actor net[, critic net]) over the N kitchens of the observation it is handed.  Keeps the last log-probabilities and
    values in `.action_logp` / `.values` ([N] float32) for a runner that records them."""

    def __init__(self, nets: PackedNetworks, n_kitchens: int, segments, deterministic: bool = False, seed: int = 0):
        rows = []
        self.has_critic = False
        for seg in segments:
            first, count, actor = int(seg[0]), int(seg[1]), int(seg[2])
            rows.append((first, count, actor, TILE_ACT))
            if len(seg) > 3 and seg[3] is not None:
                rows.append((first, count, int(seg[3]), 0))
                self.has_critic = True
        self.plan = PolicyPlan(nets, int(n_kitchens), rows)
        dev = nets.device
        self.deterministic, self.seed = deterministic, seed
        self.actions = torch.zeros(int(n_kitchens), dtype=torch.int32, device=dev)
        self.action_logp = torch.zeros(int(n_kitchens), dtype=torch.float32, device=dev)
        self.values = torch.zeros(int(n_kitchens), dtype=torch.float32, device=dev) if self.has_critic else None

    def get_action(self, obs, record: bool = True) -> torch.Tensor:
        o = obs.obs if obs.obs.is_contiguous() else obs.obs.contiguous()
        self.plan.forward(o, raw=self.values, actions=self.actions, action_logp=self.action_logp,
                          deterministic=self.deterministic, seed=self.seed)
        return self.actions.view(-1, 1)                      # (N, 1), the shape the reference's agents return

    def update(self, rewards, dones) -> None:
        pass


class FusedAgentPair:
    """The ego agent and the partner agent of a reference-style loop evaluated in ONE policy launch per step.

    The reference's evaluation loop (train/testing.py:39-59) asks the ego for its action, then `env.step` asks the partner for its
    own: two policy evaluations of the SAME step's observations, one after the other -- at 1000 kitchens two launches whose fixed part
    (network into LDS, one 32-row block per workgroup) is most of a step.  `pair.ego.get_action(obs)` evaluates both players' rows
    -- the simulator's observation tensor is `[P, N, W, H, C]`, the two players' views are its two halves -- and keeps the partner's
    actions; `pair.partner.get_action(obs)`, called by `env.step` with the partner's view of the same step, hands them out.  Whenever
    that is not what happens (the view handed in is not a slice of a `[2, N, ...]` tensor, the partner is asked first or twice), each
    agent evaluates its own rows as a `FusedVectorAgent` does: same results as two separate agents with the pair's seed, one launch more.

    ego_segments / partner_segments: (first kitchen, kitchens, actor net[, critic net]) as for FusedVectorAgent."""

    class _Side:
        def __init__(self, pair, side):
            self._pair, self._side = pair, side

        def get_action(self, obs, record: bool = True):
            return self._pair._get_action(self._side, obs)

        def update(self, rewards, dones) -> None:
            pass

        @property
        def actions(self):
            return self._pair._act[self._side]

        @property
        def action_logp(self):
            return self._pair._alp[self._side]

        @property
        def values(self):
            return self._pair._val[self._side] if self._pair.has_critic else None

    def __init__(self, nets: PackedNetworks, n_kitchens: int, ego_segments, partner_segments, ego_ind: int = 0, deterministic: bool = False,
                 seed: int = 0):
        n = self.n = int(n_kitchens)
        self.nets, self.deterministic, self.seed, self.ego_ind = nets, deterministic, seed, int(ego_ind)
        if self.ego_ind not in (0, 1):
            raise ValueError("two players: ego_ind is 0 or 1")
        self.has_critic = False
        rows = [[], []]                                      # per PLAYER (row block of the [2, N] observation tensor)
        for side, segs in ((self.ego_ind, ego_segments), (1 - self.ego_ind, partner_segments)):
            for seg in segs:
                first, count, actor = int(seg[0]), int(seg[1]), int(seg[2])
                rows[side].append((first, count, actor, TILE_ACT))
                if len(seg) > 3 and seg[3] is not None:
                    rows[side].append((first, count, int(seg[3]), 0))
                    self.has_critic = True
        self._joint = PolicyPlan(nets, 2 * n, [(p * n + f, c, k, fl) for p in (0, 1) for f, c, k, fl in rows[p]])
        self._single = [PolicyPlan(nets, n, rows[p]) for p in (0, 1)]          # the fallback: one player's rows
        dev = nets.device
        self._actions = torch.zeros(2 * n, dtype=torch.int32, device=dev)
        self._logp = torch.zeros(2 * n, dtype=torch.float32, device=dev)
        self._values = torch.zeros(2 * n, dtype=torch.float32, device=dev) if self.has_critic else None
        self._act = [self._actions[:n], self._actions[n:]]
        self._alp = [self._logp[:n], self._logp[n:]]
        self._val = [self._values[:n], self._values[n:]] if self.has_critic else [None, None]
        self._ready, self._ready_ptr = None, 0               # player whose actions the last joint launch left to be handed out, and of which tensor
        self.ego, self.partner = self._Side(self, self.ego_ind), self._Side(self, 1 - self.ego_ind)
        self.joint_launches = self.single_launches = 0

    def _whole(self, side, view):
        """The `[2, N, ...]` tensor `view` is player `side`'s half of, or None."""
        base = view._base
        rb = self.nets.W * self.nets.H * 20
        if base is None or base.dtype != torch.int8 or not base.is_contiguous() or base.numel() != 2 * self.n * rb or view.numel() != self.n * rb:
            return None
        if not view.is_contiguous() or view.data_ptr() != base.data_ptr() + side * self.n * rb:
            return None
        return base

    def _get_action(self, side, obs):
        view = obs.obs
        if self._ready == side:                              # evaluated with the other player's rows a moment ago, on this step's observations
            self._ready = None
            w = self._whole(side, view)
            if w is not None and w.data_ptr() == self._ready_ptr:   # (the same simulator's tensor: not another env's view)
                return self._act[side].view(-1, 1)
        whole = self._whole(side, view) if side == self.ego_ind else None
        if whole is not None:
            self._joint.forward(whole, raw=self._values, actions=self._actions, action_logp=self._logp, deterministic=self.deterministic, seed=self.seed)
            self._ready, self._ready_ptr = 1 - side, whole.data_ptr()
            self.joint_launches += 1
        else:
            o = view if view.is_contiguous() else view.contiguous()
            self._single[side].forward(o, raw=self._val[side], actions=self._act[side], action_logp=self._alp[side], deterministic=self.deterministic,
                                       seed=self.seed + 1 + side)
            self._ready = None
            self.single_launches += 1
        return self._act[side].view(-1, 1)
