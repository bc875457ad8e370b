"""generate_env: the factory the reference trainers call (train/env_utils.py:10-28)."""
from .overcooked_env import OvercookedMadrona


def generate_env(name, num_envs, layout="simple", use_env_cpu=False, use_baseline=False, gpu_id=0):
    if name != "overcooked":
        raise Exception("Invalid environment name")          # same error as the reference for unknown names
    if use_baseline:
        raise RuntimeError("use_baseline=True selects the reference's Python SyncVectorEnv baseline, which is not "
                           "part of the MI355X engine (see oracle/ for the CPU checker used by the tests)")
    return OvercookedMadrona(layout, num_envs, gpu_id, debug_compile=False, use_env_cpu=use_env_cpu)
