import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tools") not in sys.path:          # tools/torch_policies.py: the plain-PyTorch fp32 reference of the fused policy kernel
    sys.path.insert(1, os.path.join(ROOT, "tools"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built libraries are git-ignored but travel to the GPU box with the snapshot: (re)build them whenever a source is
    # newer than the library -- both build() functions are mtime-gated and return at once otherwise -- so the suite never
    # validates a stale .so and does not depend on the order in which a driver runs "build" and "test".  hipcc
    # cross-compiles for gfx950 without a GPU; this is the build, not a fallback -- the product still refuses to run
    # without its library.
    from diverse_conventions_amd import native
    from oracle import oracle
    native.build()
    oracle.build()


class CleanLauncher:
    """Runs commands in a helper process that never touched the GPU (tests/_clean_launcher.py)."""

    def __init__(self):
        import subprocess
        self.proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_clean_launcher.py")], stdin=subprocess.PIPE,
                                     stdout=subprocess.PIPE, text=True, cwd=ROOT)

    def run(self, cmd, env=None, timeout=600, drop_env=()):
        self.proc.stdin.write(json.dumps({"cmd": list(cmd), "env": env or {}, "drop_env": list(drop_env), "cwd": ROOT, "timeout": timeout}) + "\n")
        self.proc.stdin.flush()
        return json.loads(self.proc.stdout.readline())

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=10)
        except Exception:
            self.proc.kill()


_launcher = None


def pytest_sessionstart(session):
    # Started here, before any test (or the collection hook below) initialises the GPU; counting devices does not.
    global _launcher
    try:
        import torch
        if torch.cuda.device_count() > 0:
            _launcher = CleanLauncher()
    except Exception:
        _launcher = None


def pytest_sessionfinish(session, exitstatus):
    if _launcher is not None:
        _launcher.close()


@pytest.fixture(scope="session")
def clean_launcher():
    if _launcher is None:
        pytest.skip("no GPU box: the clean launcher is only started there")
    return _launcher


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def golden_layout_of(path, rec):
    if "layout" in rec:
        return str(rec["layout"])
    base = os.path.basename(path)[:-4]
    for prefix in ("random_", "guided_"):
        if base.startswith(prefix):
            return base[len(prefix):]
    if base == "scripted_recipe_simple":
        return "simple"
    raise KeyError(path)


@pytest.fixture(scope="session")
def layout_params():
    with open(os.path.join(GOLDEN, "layout_params.json")) as f:
        return json.load(f)


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; skip them when no GPU is visible so a bare
    # `pytest tests` still passes in the CPU container.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
