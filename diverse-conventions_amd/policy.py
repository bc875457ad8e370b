"""Segment-batched policy inference, PyTorch-ROCm as the north star prescribes for the
small policy networks.

The reference evaluates every convention's actor separately and sequentially on its env segment --
2(2k+1) tiny CNN forwards of several kernel launches each per env step (train/XD/xd_player.py:190-207,
train/partner_agents.py:97-111).  `StackedActors` keeps K actors of the reference architecture
(R_Actor = CNNBase + Categorical head: train/MAPPO/r_actor_critic.py:11-76, utils/cnn.py:12-60,
utils/act.py, utils/distributions.py:60-80) with their weights stacked on a leading axis and evaluates all
segments with four batched GEMMs, reading the engine's int8 observations directly.
"""
from typing import List, Sequence

import torch


class StackedActors:
    """K reference-format actor state dicts -> batched evaluation.

    obs int8/float [G, B, W, H, C] (G segments of B kitchens), policy_index [G] (which actor plays segment g)
    -> normalised action log-probabilities [G, B, A]."""

    def __init__(self, state_dicts: Sequence[dict], obs_shape, device, dtype=torch.float32):
        self.W, self.H, self.C = (int(x) for x in obs_shape)
        self.device, self.dtype = torch.device(device), dtype
        t = lambda sd, k: torch.as_tensor(sd[k]).to(device=self.device, dtype=dtype)
        conv_w = torch.stack([t(sd, "base.cnn.cnn.0.weight") for sd in state_dicts])      # [K, O, C, 3, 3]
        K, O, C, kw, kh = conv_w.shape
        assert (C, kw, kh) == (self.C, 3, 3)
        self.K, self.O = K, O
        self.P = (self.W - 2) * (self.H - 2)                                              # conv output positions
        # conv as a GEMM on 3x3 patches: rows ordered (c, kx, ky) like weight.reshape(O, C*9)
        self.w_conv = conv_w.reshape(K, O, C * 9).transpose(1, 2).contiguous()            # [K, C*9, O]
        self.b_conv = torch.stack([t(sd, "base.cnn.cnn.0.bias") for sd in state_dicts])   # [K, O]
        fc1 = torch.stack([t(sd, "base.cnn.cnn.3.weight") for sd in state_dicts])         # [K, Hd, O*P], cols (o, p)
        Hd = fc1.shape[1]
        # our conv output is position-major (p, o); permute the columns once instead of transposing activations
        self.w_fc1 = fc1.reshape(K, Hd, O, self.P).permute(0, 3, 2, 1).reshape(K, self.P * O, Hd).contiguous()
        self.b_fc1 = torch.stack([t(sd, "base.cnn.cnn.3.bias") for sd in state_dicts])
        self.w_fc2 = torch.stack([t(sd, "base.cnn.cnn.5.weight") for sd in state_dicts]).transpose(1, 2).contiguous()
        self.b_fc2 = torch.stack([t(sd, "base.cnn.cnn.5.bias") for sd in state_dicts])
        self.w_act = torch.stack([t(sd, "act.action_out.linear.weight") for sd in state_dicts]).transpose(1, 2).contiguous()
        self.b_act = torch.stack([t(sd, "act.action_out.linear.bias") for sd in state_dicts])

    def log_probs(self, obs: torch.Tensor, policy_index: torch.Tensor) -> torch.Tensor:
        G, B = obs.shape[0], obs.shape[1]
        idx = policy_index.to(self.device)
        x = obs.to(self.dtype)                                                            # int8 -> float once
        # [G, B, W-2, H-2, C, 3, 3] windows -> [G, B*P, C*9]
        x = x.unfold(2, 3, 1).unfold(3, 3, 1).reshape(G, B * self.P, self.C * 9)
        x = torch.relu(torch.baddbmm(self.b_conv[idx].unsqueeze(1), x, self.w_conv[idx]))         # [G, B*P, O]
        x = x.reshape(G, B, self.P * self.O)
        x = torch.relu(torch.baddbmm(self.b_fc1[idx].unsqueeze(1), x, self.w_fc1[idx]))
        x = torch.relu(torch.baddbmm(self.b_fc2[idx].unsqueeze(1), x, self.w_fc2[idx]))
        logits = torch.baddbmm(self.b_act[idx].unsqueeze(1), x, self.w_act[idx])
        return torch.log_softmax(logits, dim=-1)                                           # Categorical(logits=).logits

    def act(self, obs, policy_index, deterministic=False, generator=None):
        """-> actions int64 [G, B, 1], their log-probs [G, B, 1] (reference FixedCategorical.sample / .mode)."""
        logp = self.log_probs(obs, policy_index)
        if deterministic:
            a = logp.argmax(dim=-1, keepdim=True)
        else:
            # Gumbel-max: one fused elementwise kernel + argmax, capturable in a HIP graph
            u = torch.rand(logp.shape, device=logp.device, dtype=logp.dtype, generator=generator).clamp_(1e-12, 1.0)
            a = (logp - torch.log(-torch.log(u))).argmax(dim=-1, keepdim=True)
        return a, logp.gather(-1, a)


def segment_policy_indices(segments, side: str) -> List[int]:
    """Policy index per pairing segment for 'ego' or 'partner' (see sharding.pairing_segments)."""
    return [s.ego_policy if side == "ego" else s.partner_policy for s in segments]
