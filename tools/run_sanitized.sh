#!/bin/bash
# CPU-only ASan + UBSan run of the product library's HOST code (tile / group planners, hiprtc source generation and the on-disk
# code-object cache, argument checks, the C-ABI error paths) through the tests that need no GPU.  Device code is not instrumented
# (GPU sanitizers need xnack+ code objects; the GPU pool refuses them), so this is the sanitizer coverage the product gets.
#   tools/run_sanitized.sh [extra pytest args]
set -e
cd "$(dirname "$0")/.."
ASAN_RT=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so)
CACHE=$(mktemp -d)
trap 'rm -rf "$CACHE"' EXIT
export OC_SANITIZE=1 OC_JIT_CACHE="$CACHE"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
# (of test_abi's per-layout compile checks a handful: each is the same host code with another table; the cache test starts
# uninstrumented child interpreters and is left to the plain suite)
LD_PRELOAD="$ASAN_RT" python -m pytest tests/test_abi.py tests/test_group.py tests/test_fused_policy.py tests/test_host_logic.py tests/test_layouts.py \
    -m "not gpu" -q -p no:cacheprovider \
    -k "not code_object_cache and (not per_engine_kernel_build or simple or unident_s or six_pots or wide_hall or random3)" "$@"
# a second process: the code objects the run above wrote are now read back from the disk cache (the read path), and the
# process says which library it really mapped
LD_PRELOAD="$ASAN_RT" python - <<'PY'
from diverse_conventions_amd import get_base_layout_params, native
assert native.SANITIZE and native.LIB_PATH.endswith("liboc_hip_asan.so")
for name in ("simple", "unident_s"):
    assert native.jit_compile_check(get_base_layout_params(name, 200)) > 10_000
maps = open("/proc/self/maps").read()
assert "liboc_hip_asan.so" in maps and "libclang_rt.asan" in maps and "/liboc_hip.so" not in maps
print("sanitized library mapped: liboc_hip_asan.so + libclang_rt.asan; disk-cache read path exercised")
PY
