"""The PyTorch-ROCm batched-GEMM evaluation of the reference's policy networks and the rollout step built on it: the COMPARISON
BASELINE of SURVEY.md 8(f2), not a product path (round 6: moved out of the package -- the product evaluates policies with the fused
MFMA kernel, csrc/policy_ops.hip, 48 us per step where this path takes 684).  tests/ use it as the plain-PyTorch fp32 reference the
fused kernel is compared with; tools/bench_fused_policy.py and bench_paired_rollout.py time it beside the kernel.

The reference evaluates every convention's actor separately and sequentially on its env segment -- 2(2k+1) tiny CNN forwards of
several kernel launches each per env step (train/XD/xd_player.py:190-207, train/partner_agents.py:97-111).  `StackedActors` keeps K
actors of the reference architecture (R_Actor = CNNBase + Categorical head: train/MAPPO/r_actor_critic.py:11-76, utils/cnn.py:12-60,
utils/act.py, utils/distributions.py:60-80) with their weights stacked on a leading axis and evaluates all segments with four batched
GEMMs, reading the engine's int8 observations directly (the 3x3 convolution is expanded once into a dense matrix over the flattened
observation row, so there is no im2col pass).
"""
import os
import sys
from typing import Sequence

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from diverse_conventions_amd.engine import OvercookedEngine                       # noqa: E402
from diverse_conventions_amd.sharding import pairing_segments, segment_policy_indices   # noqa: E402


class StackedActors:
    """K reference-format actor state dicts -> batched evaluation.

    obs int8/float [G, B, W, H, C] (G segments of B kitchens), policy_index [G] (which actor plays segment g)
    -> normalised action log-probabilities [G, B, A]."""

    HEAD = "act.action_out.linear"          # final Linear of the network (reference utils/distributions.py:60-80)

    def __init__(self, state_dicts: Sequence[dict], obs_shape, device, dtype=torch.float32):
        self.W, self.H, self.C = (int(x) for x in obs_shape)
        self.device, self.dtype = torch.device(device), dtype
        t = lambda sd, k: torch.as_tensor(sd[k]).to(device=self.device, dtype=dtype)
        conv_w = torch.stack([t(sd, "base.cnn.cnn.0.weight") for sd in state_dicts])      # [K, O, C, 3, 3]
        K, O, C, kw, kh = conv_w.shape
        assert (C, kw, kh) == (self.C, 3, 3)
        self.K, self.O = K, O
        self.P = (self.W - 2) * (self.H - 2)                                              # conv output positions
        # The 3x3 valid convolution of a W x H x C grid is a (structured-sparse) linear map from the flattened
        # observation row (W*H*C values, exactly the engine's memory layout) to (position, channel) outputs.
        # Expanding it once into a dense [W*H*C, P*O] matrix turns the conv into ONE GEMM on the raw rows:
        # no im2col / unfold (which writes and re-reads 9x the observation bytes), ~2x the MACs, far less traffic.
        dense = torch.zeros((K, self.W, self.H, C, self.W - 2, self.H - 2, O), device=self.device, dtype=dtype)
        for x in range(self.W - 2):
            for y in range(self.H - 2):
                # conv_w[k, o, c, kx, ky] multiplies input tile (x+kx, y+ky), channel c
                dense[:, x:x + 3, y:y + 3, :, x, y, :] = conv_w.permute(0, 3, 4, 2, 1)
        self.w_conv = dense.reshape(K, self.W * self.H * C, self.P * O).contiguous()      # [K, W*H*C, P*O], cols (p, o)
        b = torch.stack([t(sd, "base.cnn.cnn.0.bias") for sd in state_dicts])             # [K, O]
        self.b_conv = b.unsqueeze(1).expand(K, self.P, O).reshape(K, self.P * O).contiguous()
        fc1 = torch.stack([t(sd, "base.cnn.cnn.3.weight") for sd in state_dicts])         # [K, Hd, O*P], cols (o, p)
        Hd = fc1.shape[1]
        # our conv output is position-major (p, o); permute the columns once instead of transposing activations
        self.w_fc1 = fc1.reshape(K, Hd, O, self.P).permute(0, 3, 2, 1).reshape(K, self.P * O, Hd).contiguous()
        self.b_fc1 = torch.stack([t(sd, "base.cnn.cnn.3.bias") for sd in state_dicts])
        self.w_fc2 = torch.stack([t(sd, "base.cnn.cnn.5.weight") for sd in state_dicts]).transpose(1, 2).contiguous()
        self.b_fc2 = torch.stack([t(sd, "base.cnn.cnn.5.bias") for sd in state_dicts])
        self.w_act = torch.stack([t(sd, self.HEAD + ".weight") for sd in state_dicts]).transpose(1, 2).contiguous()
        self.b_act = torch.stack([t(sd, self.HEAD + ".bias") for sd in state_dicts])

    def select(self, policy_index: torch.Tensor):
        """Per-segment weight stacks for a fixed segment->policy map, gathered once (the map changes only when a
        new convention starts training) instead of once per env step."""
        idx = policy_index.to(self.device)
        return tuple(w[idx].contiguous() for w in (self.w_conv, self.w_fc1, self.w_fc2, self.w_act)) + \
            tuple(b[idx].unsqueeze(1).contiguous() for b in (self.b_conv, self.b_fc1, self.b_fc2, self.b_act))

    def head_outputs(self, obs: torch.Tensor, policy_index=None, selected=None) -> torch.Tensor:
        """Raw outputs of the final Linear: action logits for actors, values for critics.  [G, B, out]"""
        G, B = obs.shape[0], obs.shape[1]
        wc, w1, w2, wa, bc, b1, b2, ba = selected if selected is not None else self.select(policy_index)
        x = obs.reshape(G, B, self.W * self.H * self.C).to(self.dtype)                    # int8 -> float once
        x = torch.relu(torch.baddbmm(bc, x, wc))                                           # [G, B, P*O]
        x = torch.relu(torch.baddbmm(b1, x, w1))
        x = torch.relu(torch.baddbmm(b2, x, w2))
        return torch.baddbmm(ba, x, wa)

    def log_probs(self, obs: torch.Tensor, policy_index=None, selected=None) -> torch.Tensor:
        return torch.log_softmax(self.head_outputs(obs, policy_index, selected), dim=-1)     # Categorical(logits=).logits

    def act(self, obs, policy_index=None, deterministic=False, generator=None, selected=None):
        """-> actions int64 [G, B, 1], their log-probs [G, B, 1] (reference FixedCategorical.sample / .mode)."""
        logp = self.log_probs(obs, policy_index, selected)
        if deterministic:
            a = logp.argmax(dim=-1, keepdim=True)
        else:
            # Gumbel-max: one fused elementwise kernel + argmax, capturable in a HIP graph
            # u drawn in f32 and kept inside (0, 1): u == 1 (reachable when a low-precision draw rounds up) would make
            # the noise +inf and force that action whatever its probability
            u = torch.rand(logp.shape, device=logp.device, dtype=torch.float32, generator=generator).clamp_(2.0 ** -24, 1.0 - 2.0 ** -24)
            a = (logp.float() - torch.log(-torch.log(u))).argmax(dim=-1, keepdim=True)
        return a, logp.gather(-1, a)


class StackedCritics(StackedActors):
    """K reference-format critics (R_Critic = CNNBase + `v_out` Linear, train/MAPPO/r_actor_critic.py:139-199;
    the sp / xp0[i] / xp1[i] / mp critics of MCPolicy, train/XD/MCPolicy.py:7-54), evaluated per segment in the
    same four batched GEMMs.  `values(obs, critic_index)` -> [G, B, 1]."""

    HEAD = "v_out"

    def values(self, obs, critic_index=None, selected=None):
        return self.head_outputs(obs, critic_index, selected)


class PairedRollout:
    def __init__(self, engine: OvercookedEngine, actors: StackedActors, threads: int, n_conventions: int,
                 ego_id: int = 0, deterministic: bool = False):
        self.engine, self.actors, self.threads = engine, actors, threads
        self.segments = pairing_segments(threads, n_conventions)
        g = len(self.segments)
        if engine.num_kitchens != g * threads:
            raise ValueError(f"engine must hold threads*(2k+1) = {g * threads} kitchens")
        dev = engine.device
        self.ego_id = ego_id
        self.idx = [None, None]
        self.idx[ego_id] = torch.tensor(segment_policy_indices(self.segments, "ego"), device=dev)
        self.idx[1 - ego_id] = torch.tensor(segment_policy_indices(self.segments, "partner"), device=dev)
        self.selected = [actors.select(i) for i in self.idx]     # gathered once per pairing
        self.deterministic = deterministic
        self.actions = torch.zeros((2, engine.num_kitchens), dtype=torch.int64, device=dev)
        self.logp = torch.zeros((2, engine.num_kitchens), dtype=actors.dtype, device=dev)
        self.graph = None

    def _step(self):
        e = self.engine
        g = len(self.segments)
        for p in range(2):                                   # both sides, every segment, four GEMMs each
            obs = e.obs[p].view(g, self.threads, e.width, e.height, e.obs.shape[-1])
            a, lp = self.actors.act(obs, deterministic=self.deterministic, selected=self.selected[p])
            self.actions[p].copy_(a.view(-1))
            self.logp[p].copy_(lp.view(-1))
        e.step(self.actions)                                 # int64 actions are read as they are

    def step(self):
        """One env step for all segments -> (obs, reward, done) engine buffers (overwritten next step)."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step()
        return self.engine.obs, self.engine.reward, self.engine.done

    def capture(self, warmup: int = 3):
        """Record policy GEMMs + sampling + env kernel once; replay costs one graph launch per step."""
        self.engine.prepare_kernels()                         # nothing may compile or load a module inside the capture
        s = torch.cuda.Stream(self.engine.device)
        s.wait_stream(torch.cuda.current_stream(self.engine.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream(self.engine.device).wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        return self
