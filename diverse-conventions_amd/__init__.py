"""MI355X-native Overcooked rollout engine behind the reference's vector-env API.

    from diverse_conventions_amd import OvercookedMadrona, generate_env
"""
from .layouts import ALIASES, CLASSIC_LAYOUTS, get_base_layout_params, layout_names
from .vector_api import PlayerException, RandomVectorAgent, VectorAgent, VectorMultiAgentEnv, VectorObservation
from .engine import OvercookedEngine
from .group import EngineGroup, GroupPairedRollout
from .overcooked_env import OvercookedMadrona
from .env_utils import generate_env
from .trajectory import TrajectoryBuffer
from .rollout_ops import gae_returns, gather_columns, gather_rows, normalize_advantages
from .minibatch import MinibatchSource
from .rollout import FusedPairedRollout
from .fused_policy import FusedAgentPair, FusedVectorAgent, PackedNetworks, PolicyPlan
from .mixed_play import MixedPlayRollout, MixedPlaySchedule
from .sharding import ChunkedGather, KitchenShard, Segment, TrajectoryExchange, aligned_threads, local_segments, pairing_segments, segment_critics, segment_policy_indices

__all__ = [
    "ALIASES", "CLASSIC_LAYOUTS", "get_base_layout_params", "layout_names", "RandomVectorAgent", "VectorAgent",
    "PlayerException", "VectorMultiAgentEnv", "VectorObservation", "OvercookedEngine", "EngineGroup", "GroupPairedRollout", "OvercookedMadrona",
    "generate_env", "TrajectoryBuffer", "gae_returns", "normalize_advantages", "gather_rows", "gather_columns", "MinibatchSource", "FusedPairedRollout", "PackedNetworks", "PolicyPlan", "FusedVectorAgent", "FusedAgentPair", "MixedPlaySchedule", "MixedPlayRollout", "KitchenShard", "ChunkedGather", "Segment", "TrajectoryExchange", "local_segments", "pairing_segments", "segment_policy_indices", "segment_critics", "aligned_threads",
]
