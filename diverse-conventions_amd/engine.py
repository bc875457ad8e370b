"""OvercookedEngine: thin Python owner of one native engine handle plus the torch tensors the
kernels read and write.  PyTorch is plumbing here (device memory, streams); all simulation and
featurisation happens in csrc/overcooked_engine.hip through the C ABI.
"""
import ctypes

import numpy as np
import torch

from . import native

NUM_ACTIONS = 6
CHANNELS = 20


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


try:                                    # the current stream's raw handle without building a torch.cuda.Stream object
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                  # pragma: no cover - older torch
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


def pack_states(tiles, players, timestep, record_bytes):
    """Fixture serialisation (tiles int8 [N,S,4], players int8 [N,P,6], timestep int32 [N]) ->
    native state records uint32 [N, record_bytes/4] (layouts: include/overcooked_hip.h; P = 2 packs both players'
    position/orientation into one word, other player counts use one word per player)."""
    tiles = np.ascontiguousarray(tiles, dtype=np.int8)
    players = np.ascontiguousarray(players, dtype=np.int8)
    n, s, P = tiles.shape[0], tiles.shape[1], players.shape[1]
    rec = np.zeros((n, record_bytes // 4), dtype=np.uint32)
    rec[:, :s] = tiles.view(np.uint8).view("<u4").reshape(n, s)
    pu = players.view(np.uint8).astype(np.uint32)
    held = lambda i: pu[:, i, 2] | (pu[:, i, 3] << 8) | (pu[:, i, 4] << 16) | (pu[:, i, 5] << 24)
    if P == 2:
        rec[:, s] = pu[:, 0, 0] | (pu[:, 0, 1] << 8) | (pu[:, 1, 0] << 16) | (pu[:, 1, 1] << 24)
        for i in range(2):
            rec[:, s + 1 + i] = held(i)
        rec[:, s + 3] = np.asarray(timestep, dtype=np.int32).astype(np.uint32)
    else:
        for i in range(P):
            rec[:, s + i] = pu[:, i, 0] | (pu[:, i, 1] << 8)
            rec[:, s + P + i] = held(i)
        rec[:, s + 2 * P] = np.asarray(timestep, dtype=np.int32).astype(np.uint32)
    return rec


def unpack_states(rec, s, num_players=2):
    """Inverse of pack_states."""
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    n, P = rec.shape[0], num_players
    tiles = rec[:, :s].copy().view(np.uint8).reshape(n, s, 4).view(np.int8)
    players = np.zeros((n, P, 6), dtype=np.uint8)
    if P == 2:
        posor = rec[:, s]
        players[:, 0, 0] = posor & 0xFF
        players[:, 0, 1] = (posor >> 8) & 0xFF
        players[:, 1, 0] = (posor >> 16) & 0xFF
        players[:, 1, 1] = (posor >> 24) & 0xFF
        held_at, ts_at = s + 1, s + 3
    else:
        for i in range(P):
            players[:, i, 0] = rec[:, s + i] & 0xFF
            players[:, i, 1] = (rec[:, s + i] >> 8) & 0xFF
        held_at, ts_at = s + P, s + 2 * P
    for i in range(P):
        held = rec[:, held_at + i]
        for b in range(4):
            players[:, i, 2 + b] = (held >> (8 * b)) & 0xFF
    return tiles, players.view(np.int8), rec[:, ts_at].astype(np.int32)


class OvercookedEngine:
    """N kitchens of one layout on one GPU.

    Buffers (torch, on the engine's device; overwritten by every step):
      obs    int8  [P, N, W, H, C]   reward int32 [P, N]   done int32 [N]
    """

    def __init__(self, params, num_kitchens, device_index=0):
        self._handle = None
        L = native.lib()                    # raises if the HIP library is not built
        if not torch.cuda.is_available():
            raise native.OvercookedNativeError(-4, "no GPU visible to PyTorch-ROCm; the engine has no CPU fallback")
        self.params = dict(params)
        self.num_kitchens = int(num_kitchens)
        self.width, self.height = params["width"], params["height"]
        self.size = self.width * self.height
        self.num_players = params["num_players"]
        self.channels = 5 * self.num_players + 10        # 20 for the 2-player kitchens every tuned path is built for
        self.device = torch.device("cuda", device_index)
        handle = ctypes.c_void_p()
        native.check(L.oc_create(ctypes.byref(native.make_params(params)), self.num_kitchens, device_index,
                                 ctypes.byref(handle)))
        self._handle = handle
        self._L = L
        self._players_argv = None
        self.record_bytes = int(L.oc_state_record_bytes(handle))
        n = self.num_kitchens
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((self.num_players, n, self.width, self.height, self.channels), dtype=torch.int8, device=self.device)
            self.reward = torch.zeros((self.num_players, n), dtype=torch.int32, device=self.device)
            self.done = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self.observe()

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if self._handle is not None:
            torch.cuda.synchronize(self.device)
            self._L.oc_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        # (the raw handle of the current stream: torch.cuda.current_stream(...).cuda_stream builds a Stream object first, 1.9 us of
        # the 7.4 us a Python-driven step spent in this module, tools/exp_drop_in_breakdown.py)
        return _raw_stream(self.device.index)

    # ------------------------------------------------------------------ hot path
    def _actions(self, actions):
        a = actions
        if a.device != self.device:
            a = a.to(self.device)
        if a.dtype not in (torch.int32, torch.int64):
            a = a.to(torch.int32)
        if a.numel() != self.num_players * self.num_kitchens:
            raise ValueError(f"actions must have {self.num_players * self.num_kitchens} elements ([P, N] or [P, N, 1]), got {tuple(a.shape)}")
        if not a.is_contiguous():
            a = a.contiguous()
        return a, (native.OC_ACT_I64 if a.dtype == torch.int64 else native.OC_ACT_I32)

    def _records_arg(self, records, steps=None):
        """-> (pointer or None, bytes between consecutive steps' blocks) of a record destination: int32 [N, rec/4] for one step,
        [steps, N, rec/4] for a rollout; written by the step launches themselves (oc_step's d_records)."""
        if records is None:
            return None, 0
        per_step = self.num_kitchens * self.record_bytes
        want = per_step if steps is None else steps * per_step
        if (records.dtype != torch.int32 or not records.is_contiguous() or records.device != self.device
                or records.numel() * 4 != want or records.data_ptr() % 16):
            raise ValueError(f"records must be a contiguous, 16-byte aligned int32 tensor of {want} bytes on {self.device}")
        return _ptr(records), per_step

    def step(self, actions, obs=None, reward=None, done=None, records=None):
        """One fused transition+featurisation launch on the current torch stream (no host sync).
        actions: int32/int64 [P, N] or [P, N, 1].  Outputs go to the engine's own buffers unless
        explicit destination tensors (same shapes/dtypes) are given.  records: optional int32 [N, rec/4] -- the same launch
        also writes the post-step state records there (the trajectory's compact record slot; no export_states pass)."""
        a, dt = self._actions(actions)
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        done = self.done if done is None else done
        rec, _ = self._records_arg(records)
        native.check(self._L.oc_step(self._handle, _ptr(a), dt, _ptr(obs), _ptr(reward), _ptr(done), rec, self._stream()))
        return obs, reward, done

    def step_players(self, per_player):
        """One step from the two players' action tensors as they are -- `per_player` = [player 0's, player 1's], each N int32 / int64
        values on the engine's device, same dtype, contiguous -- into the engine's own buffers, without assembling a [P, N] tensor
        (oc_step_players).  -> True, or False when this call cannot be served that way (the caller then stacks and calls `step`)."""
        if self.num_players != 2:
            return False
        a0, a1 = per_player
        n = self.num_kitchens
        if (a0.dtype != a1.dtype or a0.dtype not in (torch.int32, torch.int64) or a0.device != self.device or a1.device != self.device
                or a0.numel() != n or a1.numel() != n or not a0.is_contiguous() or not a1.is_contiguous()):
            return False
        if self._players_argv is None:
            self._players_argv = (ctypes.c_void_p * 2)()
            self._own_ptrs = (self.obs.data_ptr(), self.reward.data_ptr(), self.done.data_ptr())
        self._players_argv[0], self._players_argv[1] = a0.data_ptr(), a1.data_ptr()
        rc = self._L.oc_step_players(self._handle, self._players_argv, native.OC_ACT_I64 if a0.dtype == torch.int64 else native.OC_ACT_I32,
                                     self._own_ptrs[0], self._own_ptrs[1], self._own_ptrs[2], _raw_stream(self.device.index))
        if rc == -2:                    # OC_ERR_UNSUPPORTED: a fresh destination before, another kernel build ...
            return False
        native.check(rc)
        return True

    def observe(self, obs=None):
        obs = self.obs if obs is None else obs
        native.check(self._L.oc_observe(self._handle, _ptr(obs), self._stream()))
        return obs

    def reset_all(self, obs=None):
        obs = self.obs if obs is None else obs
        native.check(self._L.oc_reset_all(self._handle, _ptr(obs), self._stream()))
        return obs

    def episode_control(self, reset_mask=None, timesteps=None):
        """Device-side per-kitchen control: kitchens with reset_mask != 0 go back to the start state, then
        timestep[k] := timesteps[k] (int32 [N] tensors on the engine's device, either may be None)."""
        def prep(t):
            if t is None:
                return None
            t = t.to(device=self.device, dtype=torch.int32).contiguous()
            if t.numel() != self.num_kitchens:
                raise ValueError("per-kitchen control vectors must have one entry per kitchen")
            return t
        m, ts = prep(reset_mask), prep(timesteps)
        native.check(self._L.oc_episode_control(self._handle, _ptr(m) if m is not None else None,
                                                _ptr(ts) if ts is not None else None, self._stream()))

    def stagger(self, period=None):
        """Spread episode phases evenly: kitchen k starts at timestep k*period//N, so resets arrive every few
        steps instead of all at once (the diagonal schedule of the reference's mixed-play env, generalised)."""
        period = self.params["horizon"] if period is None else period
        ts = (torch.arange(self.num_kitchens, device=self.device, dtype=torch.int64) * period // self.num_kitchens)
        self.episode_control(timesteps=ts.to(torch.int32))
        return ts

    def rollout(self, actions, steps, obs=None, reward=None, done=None, records=None):
        """`steps` consecutive launches enqueued from C.  actions [T_act, P, N] are cycled.
        obs/reward/done may carry a leading time axis of length `steps` (time-major trajectory
        buffers, written in place) or be single-step buffers that are overwritten.  records: optional int32
        [steps, N, rec/4], step t's post-step records written by step t's launch."""
        if actions.dim() < 3 or actions[0].numel() != self.num_players * self.num_kitchens:
            raise ValueError(f"rollout actions must be [T_act, P, N] with P*N = {self.num_players * self.num_kitchens}")
        t_act = actions.shape[0]
        flat = actions if actions.device == self.device else actions.to(self.device)
        if flat.dtype not in (torch.int32, torch.int64):
            flat = flat.to(torch.int32)
        flat = flat.contiguous()
        dt = native.OC_ACT_I64 if flat.dtype == torch.int64 else native.OC_ACT_I32
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        done = self.done if done is None else done
        so = obs[0].numel() if obs.dim() == 6 else 0
        sr = reward[0].numel() if reward.dim() == 3 else 0
        sd = done[0].numel() if done.dim() == 2 else 0
        rec, rec_stride = self._records_arg(records, steps)
        native.check(self._L.oc_rollout(self._handle, _ptr(flat), dt, t_act, steps, _ptr(obs), so, _ptr(reward), sr,
                                        _ptr(done), sd, rec, rec_stride, self._stream()))
        return obs, reward, done

    def rollout_fused(self, steps, actions=None, seed=0, t0=0, obs=None, reward=None, done=None, records=None):
        """The whole rollout in one launch (state stays on chip between steps).  actions: tape [T_act, P, N]
        used cyclically, or None for the built-in counter-hash stream (`synthetic_actions` reproduces it).
        obs/reward/done may carry a leading time axis of length `steps` (filled slot by slot) or be single-step
        buffers (overwritten every step)."""
        flat, dt, t_act = None, native.OC_ACT_I32, 1
        if actions is not None:
            flat = actions if actions.device == self.device else actions.to(self.device)
            if flat.dtype not in (torch.int32, torch.int64):
                flat = flat.to(torch.int32)
            flat = flat.contiguous()
            if flat.dim() < 3 or flat[0].numel() != self.num_players * self.num_kitchens:
                raise ValueError("rollout actions must be [T_act, P, N]")
            dt, t_act = (native.OC_ACT_I64 if flat.dtype == torch.int64 else native.OC_ACT_I32), flat.shape[0]
        obs = self.obs if obs is None else obs
        reward = self.reward if reward is None else reward
        done = self.done if done is None else done
        so = obs[0].numel() if obs.dim() == 6 else 0
        sr = reward[0].numel() if reward.dim() == 3 else 0
        sd = done[0].numel() if done.dim() == 2 else 0
        rec, rec_stride = self._records_arg(records, steps)
        native.check(self._L.oc_rollout_fused(self._handle, _ptr(flat) if flat is not None else None, dt, t_act, steps,
                                              seed & 0xFFFFFFFF, t0 & 0xFFFFFFFF, _ptr(obs), so, _ptr(reward), sr,
                                              _ptr(done), sd, rec, rec_stride, self._stream()))
        return obs, reward, done

    @staticmethod
    def synthetic_actions(seed, num_kitchens, t0, steps):
        """numpy int32 [steps, P, N]: the counter-hash stream `rollout_fused(actions=None)` plays."""
        k = np.arange(num_kitchens, dtype=np.uint64)[None, None, :]
        t = (np.arange(steps, dtype=np.uint64) + np.uint64(t0))[:, None, None]
        p = np.arange(2, dtype=np.uint64)[None, :, None]
        m = np.uint64(0xFFFFFFFF)
        x = (np.uint64(seed & 0xFFFFFFFF) ^ ((k * np.uint64(0x9E3779B1)) & m) ^ ((t * np.uint64(0x85EBCA77)) & m)
             ^ ((p * np.uint64(0xC2B2AE3D)) & m))
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & m
        x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & m
        x ^= x >> np.uint64(16)
        return ((x * np.uint64(6)) >> np.uint64(32)).astype(np.int32)

    def rollout_timed(self, actions, steps):
        """-> milliseconds between hipEvents bracketing `steps` step launches (synchronises)."""
        flat = actions.contiguous()
        dt = native.OC_ACT_I64 if flat.dtype == torch.int64 else native.OC_ACT_I32
        ms = ctypes.c_float()
        native.check(self._L.oc_rollout_timed(self._handle, _ptr(flat), dt, actions.shape[0], steps, _ptr(self.obs),
                                              _ptr(self.reward), _ptr(self.done), self._stream(), ctypes.byref(ms)))
        return ms.value

    def encode_states(self, records, obs=None):
        """records: uint8/int32 device tensor holding n native state records -> int8 [P, n, W, H, C]."""
        n = records.numel() * records.element_size() // self.record_bytes
        if obs is None:
            obs = torch.empty((self.num_players, n, self.width, self.height, self.channels), dtype=torch.int8, device=self.device)
        native.check(self._L.oc_encode_states(self._handle, _ptr(records), n, _ptr(obs), self._stream()))
        return obs

    @property
    def kernel_build_info(self):
        """'specialised for gfx950...' when the per-engine hiprtc build is in use, else why not."""
        return self._L.oc_kernel_build_info(self._handle).decode()

    def encode_states_into(self, records, obs_global, kitchen_offset):
        """Featurise `records` (n native state records) into kitchens [kitchen_offset, kitchen_offset+n) of a
        larger int8 [P, N_total, W, H, C] tensor (one launch, no staging copy)."""
        n = records.numel() * records.element_size() // self.record_bytes
        per_kitchen = self.size * self.channels
        if obs_global.dim() != 5 or not obs_global.is_contiguous() or kitchen_offset + n > obs_global.shape[1]:
            raise ValueError("obs_global must be a contiguous [P, N_total, W, H, C] tensor that holds the block")
        if obs_global.shape[0] != self.num_players or tuple(obs_global.shape[2:]) != (self.width, self.height, self.channels):
            raise ValueError(f"obs_global must be [{self.num_players}, N_total, {self.width}, {self.height}, {self.channels}]")
        dst = obs_global.data_ptr() + kitchen_offset * per_kitchen
        native.check(self._L.oc_encode_states_strided(self._handle, _ptr(records), n, ctypes.c_void_p(dst),
                                                      obs_global.shape[1] * per_kitchen, self._stream()))
        return obs_global

    def encode_trajectory_into(self, records, obs_global, kitchen_offset):
        """Featurise T blocks of n records (`records` [T, n, rec/4], block t possibly strided) into kitchens
        [kitchen_offset, kitchen_offset+n) of every time slot of an int8 [T, P, N_total, W, H, C] tensor: one launch."""
        T, n = records.shape[0], records.shape[1]
        per_kitchen = self.size * self.channels
        if (obs_global.dim() != 6 or not obs_global.is_contiguous() or obs_global.shape[0] != T or obs_global.shape[1] != self.num_players
                or kitchen_offset + n > obs_global.shape[2] or tuple(obs_global.shape[3:]) != (self.width, self.height, self.channels)):
            raise ValueError("obs_global must be a contiguous [T, P, N_total, W, H, C] tensor that holds the block")
        if records.stride(2) != 1 or records.stride(1) != records.shape[2] or records.shape[2] * records.element_size() != self.record_bytes:
            raise ValueError("every time step's records must be one contiguous [n, rec/4] block")
        n_total = obs_global.shape[2]
        dst = obs_global.data_ptr() + kitchen_offset * per_kitchen
        native.check(self._L.oc_encode_trajectory(self._handle, _ptr(records), n, T, records.stride(0) * records.element_size(),
                                                  ctypes.c_void_p(dst), n_total * per_kitchen, self.num_players * n_total * per_kitchen, self._stream()))
        return obs_global

    def prepare_kernels(self):
        """Build now what is otherwise compiled on first use (time-major step kernels, fused-rollout kernels): call before a HIP-graph
        capture that contains steps of this engine."""
        native.check(self._L.oc_prepare_kernels(self._handle))

    def set_tuning(self, kitchens_per_wave=0, passes_per_flush=0, waves_per_cu=0):
        native.check(self._L.oc_set_tuning(self._handle, kitchens_per_wave, passes_per_flush, waves_per_cu))

    # ------------------------------------------------------------------ state import/export
    def export_states(self, first=0, count=None, out=None):
        count = self.num_kitchens - first if count is None else count
        if out is None:
            out = torch.empty((count, self.record_bytes // 4), dtype=torch.int32, device=self.device)
        native.check(self._L.oc_export_states(self._handle, first, count, _ptr(out), self._stream()))
        return out

    def import_states(self, records, first=0):
        count = records.numel() * records.element_size() // self.record_bytes
        native.check(self._L.oc_import_states(self._handle, first, count, _ptr(records), self._stream()))

    def state_dict(self):
        """Checkpoint of the simulation (the reference never checkpoints env state; torch
        format like its model checkpoints): native records on the CPU plus what is needed to validate a reload."""
        return {"records": self.export_states().cpu(), "layout": {k: (list(v) if isinstance(v, (list, tuple)) else v)
                                                                   for k, v in self.params.items()},
                "num_kitchens": self.num_kitchens, "record_bytes": self.record_bytes}

    def load_state_dict(self, sd):
        if sd["num_kitchens"] != self.num_kitchens or sd["record_bytes"] != self.record_bytes:
            raise ValueError("checkpoint was taken from an engine of another size")
        if any(sd["layout"][k] != (list(v) if isinstance(v, (list, tuple)) else v) for k, v in self.params.items()):
            raise ValueError("checkpoint was taken from another layout")
        # device path, all or nothing: one upload, one validating kernel that writes nothing (oc_validate_states), one scalar
        # read-back, and the copy only when every record passed -- a bad checkpoint leaves the engine exactly as it was
        rec = sd["records"].contiguous().to(self.device, dtype=torch.int32)
        rejected = self.validate_states(rec)
        if rejected:
            raise native.OvercookedNativeError(-1, f"{rejected} checkpoint record(s) have a player off the interior grid, a bad "
                                                   "orientation or an unknown object code; the engine's state is unchanged")
        self.import_states(rec)
        self.observe()

    def validate_states(self, records) -> int:
        """Number of `records` (device tensor in the engine's record format, any player count) the checked import would refuse;
        touches nothing."""
        count = records.numel() * records.element_size() // self.record_bytes
        rejected = torch.zeros(1, dtype=torch.int32, device=self.device)
        native.check(self._L.oc_validate_states(self._handle, count, _ptr(records), _ptr(rejected), self._stream()))
        return int(rejected.item())

    def import_states_checked(self, records, first=0) -> int:
        """Device-to-device import with a per-record validity check; returns the number of rejected (not copied) records."""
        count = records.numel() * records.element_size() // self.record_bytes
        rejected = torch.zeros(1, dtype=torch.int32, device=self.device)
        native.check(self._L.oc_import_states_checked(self._handle, first, count, _ptr(records), _ptr(rejected), self._stream()))
        return int(rejected.item())

    def get_state(self, first=0, count=None):
        """-> (tiles int8 [n,S,4], players int8 [n,2,6], timestep int32 [n]); synchronises."""
        count = self.num_kitchens - first if count is None else count
        rec = np.empty((count, self.record_bytes // 4), dtype=np.uint32)
        native.check(self._L.oc_get_state(self._handle, first, count, rec.ctypes.data_as(ctypes.c_void_p)))
        return unpack_states(rec, self.size, self.num_players)

    def set_state(self, tiles, players, timestep, first=0):
        rec = pack_states(tiles, players, timestep, self.record_bytes)
        native.check(self._L.oc_set_state(self._handle, first, rec.shape[0], rec.ctypes.data_as(ctypes.c_void_p)))
