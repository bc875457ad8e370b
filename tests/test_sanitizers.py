"""The product library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available
on the pool).  tools/run_sanitized.sh builds csrc/liboc_hip_asan.so (hipcc -fsanitize=address,undefined on the host pass, device
code uninstrumented) and re-runs, in a child interpreter with the ASan runtime preloaded, the CPU tests that drive that code: the
C-ABI error paths and the C99 client (test_abi), hiprtc source generation + the per-layout compile checks + the on-disk code-object
cache (write, then read from a second process), the group planner (test_group) and the policy tile planners / host packer
(test_fused_policy)."""
import os
import subprocess

from conftest import ROOT


def test_host_code_of_the_product_library_runs_clean_under_asan_ubsan():
    import pytest
    rt = subprocess.run(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    if rt.returncode != 0 or not os.path.isabs(rt.stdout.strip()) or not os.path.exists(rt.stdout.strip()):
        pytest.skip("this toolchain ships no ASan runtime (hipcc -print-file-name=libclang_rt.asan-x86_64.so does not resolve)")
    env = {k: v for k, v in os.environ.items() if k not in ("OC_SANITIZE", "LD_PRELOAD")}
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_sanitized.sh"), "-x"], capture_output=True, text=True, timeout=1700, env=env, cwd=ROOT)
    tail = (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "sanitized library mapped" in r.stdout and " passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail
