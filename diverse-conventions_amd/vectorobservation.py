"""Batched observation record handed to agents (same fields and defaulting rule as the reference's
pantheonrl_extension/vectorobservation.py:5-32: `state` falls back to `obs`)."""
from typing import Optional

import torch


class VectorObservation:
    """active: bool [N]; obs: [N, *obs_shape]; state: [N, *state_shape]; action_mask: bool [N, A] or None."""
    __slots__ = ("active", "obs", "state", "action_mask")

    def __init__(self, active: torch.Tensor, obs: torch.Tensor, state: Optional[torch.Tensor] = None,
                 action_mask: Optional[torch.Tensor] = None):
        self.active = active
        self.obs = obs
        self.state = state if state is not None else obs
        self.action_mask = action_mask

    def __repr__(self):
        return (f"VectorObservation(active={tuple(self.active.shape)}, obs={tuple(self.obs.shape)}, "
                f"state={tuple(self.state.shape)}, action_mask="
                f"{None if self.action_mask is None else tuple(self.action_mask.shape)})")
