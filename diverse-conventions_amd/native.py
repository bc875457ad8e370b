"""ctypes binding of the C ABI in include/overcooked_hip.h + overcooked_ops.h (library: csrc/liboc_hip.so).

The library is the product: if it is missing or cannot be loaded this module raises -- there is no
Python or CPU fallback for the hot path (the CPU oracle under oracle/ is test infrastructure).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
# OC_SANITIZE=1: the ASan + UBSan build of the library's HOST code (planners, JIT source generation, cache I/O, argument checks;
# the device code is not instrumented -- GPU sanitizers need xnack+ code objects, which the pool refuses).  CPU-only use:
#   OC_SANITIZE=1 LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
#       python -m pytest tests -m "not gpu" -k "abi or plan or planner or jit"          (tools/run_sanitized.sh)
SANITIZE = os.environ.get("OC_SANITIZE", "0") not in ("", "0")
# OC_DIAGNOSTICS=1: the DIAGNOSTICS build (-DOC_DIAGNOSTICS -> csrc/liboc_hip_diag.so): phase stamps, phase ablation, the folded env
# step and every experiment knob read from the environment (OC_STORE_MODE, OC_POLICY_PLAN, OC_POLICY_FOLD ...).  tools/ set it
# before importing the package; the product library has none of that code and reads none of those variables.
DIAGNOSTICS = os.environ.get("OC_DIAGNOSTICS", "0") not in ("", "0")
# OC_LIB_PATH=<file>: load that build of the library instead (A/B measurements of two builds in one GPU call, tools/ab_builds.sh);
# build() then leaves it alone
LIB_OVERRIDE = os.environ.get("OC_LIB_PATH", "")
LIB_PATH = LIB_OVERRIDE or os.path.join(CSRC, "liboc_hip_asan.so" if SANITIZE else "liboc_hip_diag.so" if DIAGNOSTICS else "liboc_hip.so")
SOURCES = [os.path.join(CSRC, "overcooked_engine.hip"), os.path.join(CSRC, "rollout_ops.hip"),
           os.path.join(CSRC, "policy_ops.hip")]
HEADERS = [os.path.join(CSRC, "policy_pack.h"), os.path.join(CSRC, "oc_internal.h"), os.path.join(CSRC, "policy_block_body.h"), os.path.join(CSRC, "policy_mono_loop.h"),
           os.path.join(CSRC, "policy_duo_kernel.h"), os.path.join(CSRC, "policy_duo_body.h"),
           os.path.join(CSRC, "oc_np_kernels.h")]
KERNEL_HEADER = os.path.join(CSRC, "oc_step_kernel.h")              # ahead-of-time (generic) build
KERNEL_COMMON = os.path.join(CSRC, "oc_step_common.h")              # the two headers hiprtc compiles per engine / group
KERNEL_BODY = os.path.join(CSRC, "oc_step_body.h")
KERNEL_SRC_INC = os.path.join(CSRC, "oc_step_kernel_src.inc")     # generated: both headers as C++ raw strings
INCLUDE = os.path.join(REPO_ROOT, "include")

# hiprtc code objects are cached on disk (overcooked_engine.hip: jit_compile_text).  From Python the cache lives in-tree, next to
# the built library -- like liboc_hip.so it is git-ignored but travels with a snapshot of the repo, so a fresh box does not
# recompile what this checkout has compiled already; the file names hash program text, headers, target and compiler version.
os.environ.setdefault("OC_JIT_CACHE", os.path.join(REPO_ROOT, ".oc_jit_cache"))

OC_OK = 0
OC_ACT_I32, OC_ACT_I64 = 0, 1
STATUS_NAMES = {0: "OC_OK", -1: "OC_ERR_INVALID_ARG", -2: "OC_ERR_UNSUPPORTED", -3: "OC_ERR_HIP", -4: "OC_ERR_NO_DEVICE"}

# every symbol the public headers declare (tests/test_abi.py checks header text, these lists and the library's export table
# against each other): include/overcooked_hip.h -- the environment: engines, groups, state ...
SYMBOLS_ENV = [
    "oc_create", "oc_destroy", "oc_step", "oc_step_players", "oc_observe", "oc_reset_all", "oc_episode_control", "oc_rollout", "oc_rollout_timed", "oc_rollout_fused", "oc_synthetic_action",
    "oc_encode_states", "oc_encode_states_strided", "oc_encode_trajectory",
    "oc_group_create", "oc_group_destroy", "oc_group_size", "oc_group_step", "oc_group_observe", "oc_group_rollout_timed",
    "oc_group_set_tuning", "oc_group_build_info", "oc_state_record_bytes", "oc_get_state", "oc_set_state", "oc_export_states",
    "oc_import_states", "oc_import_states_checked", "oc_validate_states", "oc_num_kitchens", "oc_obs_shape", "oc_device_id", "oc_set_tuning", "oc_prepare_kernels",
    "oc_kernel_build_info", "oc_last_error", "oc_abi_version",
]
# ... include/overcooked_ops.h -- policy inference, rollouts with the policies in the loop, GAE, minibatch gathers
SYMBOLS_OPS = [
    "oc_gae_returns", "oc_advantage_workspace_bytes", "oc_advantage_normalize", "oc_gather_rows", "oc_gather_columns", "oc_pack_columns", "oc_gather_records",
    "oc_policy_packed_bytes", "oc_policy_pack_host", "oc_policy_pack", "oc_policy_plan_tiles", "oc_policy_forward", "oc_policy_forward_multi",
    "oc_policy_plan_tiles_multi", "oc_group_policy_rollout_multi", "oc_policy_uniform", "oc_policy_rollout", "oc_group_policy_rollout",
    "oc_partition_rows", "oc_policy_forward_gather", "oc_policy_rollout_mixed",
]
SYMBOLS = SYMBOLS_ENV + SYMBOLS_OPS
# include/overcooked_experiments.h (measured and not adopted; tools/ only: exported by the diagnostics build only since round 6)
# and include/overcooked_debug.h (diagnostics)
SYMBOLS_EXPERIMENTS = ["oc_exp_cu_partition_create", "oc_exp_cu_partition_destroy", "oc_exp_policy_forward_cus", "oc_exp_set_cu_budget",
                       "oc_exp_group_policy_rollout_streams"]
SYMBOLS_DEBUG = ["oc_debug_jit_compile", "oc_debug_jit_compile_group", "oc_debug_jit_builds", "oc_debug_aot_build", "oc_debug_aot_build_group"]
SYMBOLS_DIAG = ["oc_debug_trace", "oc_policy_debug_trace", "oc_debug_fold_launches"]       # exported by the diagnostics build only


class OcPolicyShape(ctypes.Structure):
    """struct oc_policy_shape"""
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("hidden", ctypes.c_int32),
                ("n_out", ctypes.c_int32), ("digits", ctypes.c_int32)]


class OcGroupPolicySlot(ctypes.Structure):
    """struct oc_group_policy_slot"""
    _fields_ = [("shape", ctypes.POINTER(OcPolicyShape)), ("d_packed", ctypes.c_void_p), ("n_nets", ctypes.c_int32),
                ("d_tiles", ctypes.c_void_p), ("n_tiles", ctypes.c_int32), ("d_obs", ctypes.c_void_p), ("d_actions", ctypes.c_void_p),
                ("d_action_logp", ctypes.c_void_p), ("d_values", ctypes.c_void_p), ("d_reward", ctypes.c_void_p), ("d_done", ctypes.c_void_p),
                ("d_records", ctypes.c_void_p)]


class OcPolicyMultiSlot(ctypes.Structure):
    """struct oc_policy_multi_slot"""
    _fields_ = [("shape", ctypes.POINTER(OcPolicyShape)), ("d_packed", ctypes.c_void_p), ("n_nets", ctypes.c_int32),
                ("d_obs", ctypes.c_void_p), ("n_rows", ctypes.c_int64), ("d_raw", ctypes.c_void_p), ("d_actions", ctypes.c_void_p),
                ("d_action_logp", ctypes.c_void_p), ("d_logp", ctypes.c_void_p), ("d_row_select", ctypes.c_void_p), ("seed", ctypes.c_uint32)]


class OcLayoutParams(ctypes.Structure):
    """struct oc_layout_params"""
    _fields_ = [
        ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("num_players", ctypes.c_int32),
        ("placement_in_pot_rew", ctypes.c_int32), ("dish_pickup_rew", ctypes.c_int32),
        ("soup_pickup_rew", ctypes.c_int32), ("horizon", ctypes.c_int32),
        ("recipe_values", ctypes.c_int32 * 16), ("recipe_times", ctypes.c_int32 * 16),
        ("start_player_x", ctypes.c_int32 * 4), ("start_player_y", ctypes.c_int32 * 4),       # OC_MAX_PLAYERS
        ("terrain", ctypes.c_int32 * 256),
    ]


class OvercookedNativeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU.  Output stays in-tree (csrc/liboc_hip.so)."""
    newest = max(os.path.getmtime(p) for p in SOURCES + HEADERS + [KERNEL_HEADER, KERNEL_COMMON, KERNEL_BODY] +
                 [os.path.join(INCLUDE, h) for h in ("overcooked_hip.h", "overcooked_ops.h", "overcooked_experiments.h", "overcooked_debug.h")])
    if LIB_OVERRIDE:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"OC_LIB_PATH={LIB_PATH} does not exist")
        return LIB_PATH
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    # the kernel headers are also embedded as text: oc_create() / oc_group_create() re-compile them with hiprtc
    with open(KERNEL_SRC_INC, "w") as f:
        f.write('// generated from oc_step_common.h / oc_step_body.h by native.build() -- do not edit\n')
        for name, path in (("oc_step_common_src", KERNEL_COMMON), ("oc_step_body_src", KERNEL_BODY)):
            with open(path) as h:
                text = h.read()
            assert ')OCKSRC"' not in text
            f.write('static const char ' + name + '[] = R"OCKSRC(' + text + ')OCKSRC";\n')
    # -ffp-contract=off: rollout_ops.hip must round like the reference's separate torch ops (no FMA); the
    # engine itself has no floating point
    # -amdgpu-mfma-vgpr-form: MFMA results stay in ordinary VGPRs (policy_ops.hip reads them with VALU ops; the
    # accumulator-file form cost 64 v_accvgpr_read per conv position and kept the kernel at one wavefront per SIMD)
    opt = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-shared-libsan"] if SANITIZE else ["-O3"]
    if DIAGNOSTICS:
        opt = opt + ["-DOC_DIAGNOSTICS"]
    common = ["hipcc", "--offload-arch=gfx950"] + opt + ["-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-I", INCLUDE]
    # the three translation units side by side (policy_ops.hip alone is two thirds of the time), then one link.  Objects go to a
    # directory of this process's own and the library is renamed into place whole: two builds running side by side (pytest-xdist, an
    # A/B of two checkouts sharing build/) can neither clobber each other's objects nor expose a half-written .so
    import shutil
    import tempfile
    os.makedirs(os.path.join(REPO_ROOT, "build"), exist_ok=True)
    objdir = tempfile.mkdtemp(prefix="obj_asan." if SANITIZE else "obj_diag." if DIAGNOSTICS else "obj.", dir=os.path.join(REPO_ROOT, "build"))
    try:
        objs = [os.path.join(objdir, os.path.basename(src)[:-4] + ".o") for src in SOURCES]
        # policy_ops.hip: its kernels' two leading scalar arguments (table pointer, tiles per workgroup) preloaded into SGPRs (prefetch_own_tile)
        extra = {"policy_ops.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=4"]}
        cmds = [common + extra.get(os.path.basename(src), []) + ["-c", src, "-o", obj] for src, obj in zip(SOURCES, objs)]
        if verbose:
            for c in cmds:
                print(" ".join(c))
        procs = [subprocess.Popen(c) for c in cmds]
        failed = [c for c, pr in zip(cmds, procs) if pr.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(1, failed[0])
        tmp_lib = os.path.join(objdir, os.path.basename(LIB_PATH))
        link = ["hipcc", "--offload-arch=gfx950"] + (["-fsanitize=address,undefined", "-shared-libsan"] if SANITIZE else []) + ["-shared", "-fPIC", "-o", tmp_lib] + objs + ["-lhiprtc"]
        if verbose:
            print(" ".join(link))
        subprocess.run(link, check=True)
        os.replace(tmp_lib, LIB_PATH)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP engine first "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    L.oc_create.argtypes = [ctypes.POINTER(OcLayoutParams), i64, i32, ctypes.POINTER(vp)]
    L.oc_destroy.argtypes = [vp]
    L.oc_step.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.oc_step_players.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, vp, vp]
    L.oc_observe.argtypes = [vp, vp, vp]
    L.oc_reset_all.argtypes = [vp, vp, vp]
    L.oc_episode_control.argtypes = [vp, vp, vp, vp]
    L.oc_rollout.argtypes = [vp, vp, i32, i64, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp]
    L.oc_rollout_timed.argtypes = [vp, vp, i32, i64, i64, vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    u32 = ctypes.c_uint32
    L.oc_rollout_fused.argtypes = [vp, vp, i32, i64, i64, u32, u32, vp, i64, vp, i64, vp, i64, vp, i64, vp]
    L.oc_synthetic_action.argtypes = [u32, u32, u32, u32]
    L.oc_synthetic_action.restype = u32
    L.oc_encode_states.argtypes = [vp, vp, i64, vp, vp]
    L.oc_encode_states_strided.argtypes = [vp, vp, i64, vp, i64, vp]
    L.oc_state_record_bytes.argtypes = [vp]
    L.oc_state_record_bytes.restype = i64
    L.oc_get_state.argtypes = [vp, i64, i64, vp]
    L.oc_set_state.argtypes = [vp, i64, i64, vp]
    L.oc_export_states.argtypes = [vp, i64, i64, vp, vp]
    L.oc_import_states.argtypes = [vp, i64, i64, vp, vp]
    L.oc_import_states_checked.argtypes = [vp, i64, i64, vp, vp, vp]
    L.oc_validate_states.argtypes = [vp, i64, vp, vp, vp]
    L.oc_num_kitchens.argtypes = [vp]
    L.oc_num_kitchens.restype = i64
    L.oc_obs_shape.argtypes = [vp, ctypes.POINTER(i64 * 5)]
    L.oc_device_id.argtypes = [vp]
    L.oc_set_tuning.argtypes = [vp, i32, i32, i32]
    L.oc_prepare_kernels.argtypes = [vp]
    fp = ctypes.c_float
    L.oc_gae_returns.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, ctypes.c_double, ctypes.c_double, i32, fp, fp, i32, vp]
    L.oc_advantage_workspace_bytes.argtypes = []
    L.oc_advantage_workspace_bytes.restype = i64
    L.oc_advantage_normalize.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    L.oc_gather_rows.argtypes = [vp, i64, ctypes.c_int32, vp, i64, vp, i32, vp]
    L.oc_gather_columns.argtypes = [ctypes.c_int32, ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(vp), ctypes.POINTER(vp), vp, i64, i64, i64, vp]
    L.oc_pack_columns.argtypes = [ctypes.c_int32, ctypes.POINTER(vp), i64, vp, vp]
    L.oc_gather_records.argtypes = [vp, i64, ctypes.c_int32, ctypes.POINTER(vp), ctypes.c_int32, vp, vp, i64, i64, i64, vp]
    ps = ctypes.POINTER(OcPolicyShape)
    L.oc_policy_packed_bytes.argtypes = [ps]
    L.oc_policy_packed_bytes.restype = i64
    L.oc_policy_pack_host.argtypes = [ps] + [vp] * 9
    L.oc_policy_pack.argtypes = [ps] + [vp] * 10
    L.oc_policy_plan_tiles.argtypes = [vp, vp, vp, vp, ctypes.c_int32, ctypes.c_int32, vp, i64]
    L.oc_policy_plan_tiles.restype = i64
    L.oc_policy_forward.argtypes = [ps, vp, ctypes.c_int32, vp, i64, vp, ctypes.c_int32, vp, vp, vp, vp, i32, u32, u32, vp, vp, vp]
    L.oc_partition_rows.argtypes = [vp, i64, i64, i64, vp, vp, vp]
    L.oc_policy_forward_gather.argtypes = [ps, vp, ctypes.c_int32, vp, i64, vp, ctypes.c_int32, vp, vp, i64, vp, vp, vp, vp, i32, u32, u32, vp, vp]
    L.oc_policy_rollout_mixed.argtypes = [vp, ps, vp, ctypes.c_int32, vp, ctypes.c_int32, i64, vp, vp, vp, vp, vp, vp, vp, i32, u32, u32, vp, vp, vp, vp]
    L.oc_policy_rollout.argtypes = [vp, ps, vp, ctypes.c_int32, vp, ctypes.c_int32, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, u32, u32, vp, vp]
    L.oc_group_policy_rollout.argtypes = [vp, ctypes.POINTER(OcGroupPolicySlot), i64, i32, i32, u32, u32, vp]
    if DIAGNOSTICS:
        L.oc_exp_set_cu_budget.argtypes = [vp, ctypes.c_int32]
        L.oc_exp_policy_forward_cus.argtypes = [ps, vp, ctypes.c_int32, vp, i64, vp, ctypes.c_int32, vp, vp, vp, vp, i32, u32, u32, vp, vp, ctypes.c_int32, vp]
        L.oc_exp_cu_partition_create.argtypes = [i32, ctypes.POINTER(ctypes.c_float), ctypes.c_int32, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)]
        L.oc_exp_cu_partition_destroy.argtypes = [ctypes.POINTER(vp), ctypes.c_int32]
        L.oc_exp_group_policy_rollout_streams.argtypes = [vp, ctypes.POINTER(OcGroupPolicySlot), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32), i64, i32, i32,
                                                          u32, u32, vp]
        L.oc_debug_trace.argtypes = [vp, i32, vp, i64]
        L.oc_policy_debug_trace.argtypes = [i32, vp, i64]
        L.oc_debug_fold_launches.argtypes = []
        L.oc_debug_fold_launches.restype = ctypes.c_longlong
    L.oc_group_policy_rollout_multi.argtypes = [vp, ctypes.POINTER(OcGroupPolicySlot), vp, i32, i32, i32, i64, i32, i32, u32, u32, vp]
    L.oc_policy_forward_multi.argtypes = [ctypes.POINTER(OcPolicyMultiSlot), i32, vp, i32, i32, i32, i32, u32, vp, vp]
    L.oc_policy_plan_tiles_multi.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.oc_policy_plan_tiles_multi.restype = i64
    L.oc_policy_uniform.argtypes = [u32, u32, u32, u32]
    L.oc_policy_uniform.restype = ctypes.c_float
    L.oc_last_error.argtypes = []
    L.oc_last_error.restype = ctypes.c_char_p
    L.oc_kernel_build_info.argtypes = [vp]
    L.oc_kernel_build_info.restype = ctypes.c_char_p
    L.oc_debug_jit_compile.argtypes = [ctypes.POINTER(OcLayoutParams), i64, ctypes.c_char_p, ctypes.POINTER(i64),
                                       ctypes.c_char_p, i64]
    L.oc_abi_version.argtypes = []
    L.oc_debug_jit_builds.argtypes = []
    L.oc_debug_aot_build.argtypes = [ctypes.POINTER(OcLayoutParams), i64, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32)]
    L.oc_debug_aot_build_group.argtypes = [ctypes.POINTER(OcLayoutParams), ctypes.POINTER(i64), ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, ctypes.c_char_p]
    L.oc_debug_jit_builds.restype = ctypes.c_longlong
    L.oc_encode_trajectory.argtypes = [vp, vp, i64, i64, i64, vp, i64, i64, vp]
    pvp = ctypes.POINTER(vp)
    L.oc_group_create.argtypes = [pvp, ctypes.c_int32, pvp]
    L.oc_group_destroy.argtypes = [vp]
    L.oc_group_size.argtypes = [vp]
    L.oc_group_size.restype = ctypes.c_int32
    L.oc_group_step.argtypes = [vp, pvp, i32, pvp, pvp, pvp, pvp, vp]
    L.oc_group_observe.argtypes = [vp, pvp, vp]
    L.oc_group_rollout_timed.argtypes = [vp, pvp, i32, i64, i64, pvp, pvp, pvp, vp, ctypes.POINTER(ctypes.c_float)]
    L.oc_group_set_tuning.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), i32, i32]
    L.oc_group_build_info.argtypes = [vp]
    L.oc_group_build_info.restype = ctypes.c_char_p
    L.oc_debug_jit_compile_group.argtypes = [ctypes.POINTER(OcLayoutParams), ctypes.POINTER(i64), ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_char_p, ctypes.POINTER(i64), ctypes.POINTER(ctypes.c_int32), ctypes.c_char_p, i64]
    for name in SYMBOLS + SYMBOLS_DEBUG + (SYMBOLS_EXPERIMENTS + SYMBOLS_DIAG if DIAGNOSTICS else []):
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int:          # default restype: int status
            fn.restype = ctypes.c_int
    _lib = L
    return L


def check(status):
    if status != OC_OK:
        raise OvercookedNativeError(status, lib().oc_last_error().decode())


AOT_DIR = os.path.join(CSRC, "aot")                                   # where the library looks (next to itself)
AOT_ARCH = "gfx950:sramecc+:xnack-"                                  # hipDeviceProp_t::gcnArchName of an MI355X


def aot_build(shapes=None, groups=None, arch=AOT_ARCH, out_dir=AOT_DIR, verbose=False):
    """Ahead-of-time code objects for the BASELINE shapes (no GPU needed): `shapes` = [(layout params, kitchens)], `groups` =
    [[(layout params, kitchens), ...]].  -> number of code objects in out_dir afterwards.  Defaults: see aot_default_shapes()."""
    from .layouts import get_base_layout_params
    if shapes is None or groups is None:
        d_shapes, d_groups = aot_default_shapes()
        shapes = d_shapes if shapes is None else shapes
        groups = d_groups if groups is None else groups
    L = lib()
    os.makedirs(out_dir, exist_ok=True)
    for params, n in shapes:
        cnt = ctypes.c_int32(0)
        check(L.oc_debug_aot_build(ctypes.byref(make_params(params)), int(n), arch.encode(), out_dir.encode(), ctypes.byref(cnt)))
        if verbose:
            print(f"aot: {params.get('name', '?')} x {n}: {cnt.value} modules")
    for grp in groups:
        arr = (OcLayoutParams * len(grp))(*[make_params(p) for p, _ in grp])
        counts = (ctypes.c_int64 * len(grp))(*[int(k) for _, k in grp])
        check(L.oc_debug_aot_build_group(arr, counts, len(grp), 256, arch.encode(), out_dir.encode()))
    return len([f for f in os.listdir(out_dir) if f.endswith(".hsaco")])


def aot_default_shapes():
    """What BASELINE.json's configs and bench.py's legs create: the five classic layouts at 65 536 and 8192 kitchens, cramped_room at
    4096 (config 2) and at the policies-in-the-loop shapes, config 4's per-GPU share, the config-3 group."""
    from .layouts import CLASSIC_LAYOUTS, get_base_layout_params
    P = lambda name, h=200: dict(get_base_layout_params(name, h), name=name)
    shapes = []
    for name in CLASSIC_LAYOUTS:
        shapes += [(P(name), 65536), (P(name), 8192)]
    shapes += [(P("cramped_room"), 4096), (P("cramped_room"), 65535), (P("cramped_room"), 65280), (P("cramped_room"), 1000), (P("cramped_room"), 65471),
               (P("asymmetric_advantages"), 8184), (P("asymmetric_advantages"), 65536 // 2), (P("asymmetric_advantages"), 65536 // 4)]
    groups = [[(P(name), 8192) for name in CLASSIC_LAYOUTS]]
    return shapes, groups


def jit_compile_check(params, num_kitchens=4096, arch="gfx950"):
    """Compile the per-engine (hiprtc) build of the step kernel for `params` without a GPU.
    -> code object size in bytes; raises OvercookedNativeError with the compiler log on failure."""
    size = ctypes.c_int64(0)
    log = ctypes.create_string_buffer(8192)
    check(lib().oc_debug_jit_compile(ctypes.byref(make_params(params)), num_kitchens, arch.encode(), ctypes.byref(size),
                                     log, len(log)))
    return size.value


def jit_compile_check_group(params_list, kitchens_list, num_cus=256, arch="gfx950"):
    """Compile the group module (one launch over several engines) without a GPU.
    -> (code object size, [(kitchens per wavefront group, blocks)] per engine)."""
    n = len(params_list)
    arr = (OcLayoutParams * n)(*[make_params(p) for p in params_list])
    counts = (ctypes.c_int64 * n)(*[int(k) for k in kitchens_list])
    plan = (ctypes.c_int32 * (2 * n))()
    size = ctypes.c_int64(0)
    log = ctypes.create_string_buffer(8192)
    check(lib().oc_debug_jit_compile_group(arr, counts, n, num_cus, arch.encode(), ctypes.byref(size), plan, log, len(log)))
    return size.value, [(plan[2 * i], plan[2 * i + 1]) for i in range(n)]


def make_params(params):
    """dict from get_base_layout_params -> struct oc_layout_params"""
    if not 1 <= params["num_players"] <= 4:
        raise OvercookedNativeError(-2, f"1..4 players are supported (got {params['num_players']})")
    S = params["width"] * params["height"]
    if S > 255 or len(params["terrain"]) != S:
        raise OvercookedNativeError(-2, f"grid {params['width']}x{params['height']} outside 1..255 tiles")
    p = OcLayoutParams()
    for k in ("width", "height", "num_players", "placement_in_pot_rew", "dish_pickup_rew", "soup_pickup_rew", "horizon"):
        setattr(p, k, int(params[k]))
    for i in range(16):
        p.recipe_values[i] = int(params["recipe_values"][i])
        p.recipe_times[i] = int(params["recipe_times"][i])
    for i in range(params["num_players"]):
        p.start_player_x[i] = int(params["start_player_x"][i])
        p.start_player_y[i] = int(params["start_player_y"][i])
    for i, t in enumerate(params["terrain"]):
        p.terrain[i] = int(t)
    return p
