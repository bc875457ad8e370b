"""Agent interface of the vector-env API (reference pantheonrl_extension/vectoragent.py:9-41)."""
from abc import ABC, abstractmethod

import torch

from .vectorobservation import VectorObservation


class VectorAgent(ABC):
    @abstractmethod
    def get_action(self, obs: VectorObservation, record: bool = True) -> torch.Tensor:
        """Actions [N, 1] for the N kitchens of `obs`."""

    @abstractmethod
    def update(self, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        """Rewards [N] / dones [N] that followed the most recent recorded get_action."""


class RandomVectorAgent(VectorAgent):
    """Partner that returns whatever `sampler()` produces each step."""

    def __init__(self, sampler):
        self.sampler = sampler

    def get_action(self, obs: VectorObservation, record: bool = True) -> torch.Tensor:
        return self.sampler()

    def update(self, rewards: torch.Tensor, dones: torch.Tensor) -> None:
        return
