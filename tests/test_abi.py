"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares, reports
errors by code instead of crashing, and its per-engine kernel build compiles for gfx950 (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from diverse_conventions_amd import get_base_layout_params, layout_names, native


def header_functions(name, diagnostics=False):
    """Functions a header declares; `diagnostics`: only those inside its #ifdef OC_DIAGNOSTICS block (else: only those outside)."""
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    inside = "".join(re.findall(r"#ifdef OC_DIAGNOSTICS\n(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef OC_DIAGNOSTICS\n.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oc_[a-z_0-9]+)\s*\(", inside if diagnostics else outside)))


def test_library_exports_every_declared_symbol():
    """Four headers, one library: the environment (what the reference's bindings.cpp:27-96 binds), the training-loop ops, the
    measured-and-rejected experiments (tools/ only), diagnostics.  Header text, the binding's lists and the library's dynamic
    symbol table must agree -- nothing exported that no header declares, nothing declared that is not exported."""
    import subprocess
    L = native.lib()
    assert sorted(native.SYMBOLS_ENV) == header_functions("overcooked_hip.h")
    assert sorted(native.SYMBOLS_OPS) == header_functions("overcooked_ops.h")
    assert sorted(native.SYMBOLS_EXPERIMENTS) == header_functions("overcooked_experiments.h", diagnostics=True)
    assert header_functions("overcooked_experiments.h") == []          # nothing of it outside the diagnostics build (round 6)
    assert sorted(native.SYMBOLS_DEBUG) == header_functions("overcooked_debug.h")
    assert sorted(native.SYMBOLS_DIAG) == header_functions("overcooked_debug.h", diagnostics=True)
    declared = native.SYMBOLS + native.SYMBOLS_DEBUG + (native.SYMBOLS_EXPERIMENTS + native.SYMBOLS_DIAG if native.DIAGNOSTICS else [])
    for name in declared:
        assert getattr(L, name) is not None, name
    nm = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if re.search(r" [TW] oc_[a-z_0-9]+$", ln) and "oc_internal" not in ln})
    assert exported == sorted(declared)
    assert L.oc_abi_version() == 7
    # the drop-in boundary stays small: no experiment or diagnostic leaks into the two public headers
    assert not [f for f in native.SYMBOLS if f.startswith(("oc_exp_", "oc_debug_")) or "_cus" in f or "cu_" in f]


def test_product_library_carries_no_diagnostics():
    """Result-changing timing knobs (phase ablation, the folded env step's debug modes) and lost experiments are compiled out of the
    shipped library: their environment variables do not exist in it even as strings, the phase-stamp / fold entry points are not
    exported, and no FOLD instantiation of the policy kernel is in its code objects.  (tools/ build and load csrc/liboc_hip_diag.so.)"""
    import subprocess
    if native.DIAGNOSTICS or native.SANITIZE or native.LIB_OVERRIDE:
        pytest.skip("not the product build")
    blob = open(native.LIB_PATH, "rb").read()
    for needle in (b"OC_ABLATE", b"OC_FOLD", b"OC_POLICY_FOLD", b"OC_STORE_MODE", b"OC_POLICY_PLAN", b"OC_POLICY_MONO", b"OC_JIT_TRACE", b"OC_MIN_WAVES", b"fold_env_group",
                   b"fold_expected_kernel"):
        assert needle not in blob, needle
    for kept in (b"OC_NO_JIT", b"OC_JIT_CACHE", b"OC_DWORD_STORES", b"OC_POLICY_COOP", b"OC_POLICY_DUO"):      # the documented switches of the product
        assert kept in blob, kept
    nm = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert not [s for s in native.SYMBOLS_DIAG + native.SYMBOLS_EXPERIMENTS if s in nm]
    assert b"oc_exp_" not in blob                                      # the experiments' entry points: not even as error-message strings


def test_layout_struct_matches_header():
    # 7 scalars + 2x16 recipe tables + 2x4 start coordinates (OC_MAX_PLAYERS) + 256 terrain codes, all int32
    assert ctypes.sizeof(native.OcLayoutParams) == 4 * (7 + 32 + 8 + 256)


def _create(params, n=8):
    L = native.lib()
    h = ctypes.c_void_p()
    p = native.make_params(params) if isinstance(params, dict) else params
    rc = L.oc_create(ctypes.byref(p), n, 0, ctypes.byref(h))
    if rc == 0:
        L.oc_destroy(h)
    return rc, L.oc_last_error().decode()


def test_error_codes_not_crashes():
    import torch
    L = native.lib()
    good = get_base_layout_params("simple", 200)
    rc = L.oc_create(None, 8, 0, None)
    assert rc == -1 and "NULL" in L.oc_last_error().decode()
    rc, msg = _create(good, n=0)
    assert rc == -1 and "n_kitchens" in msg
    bad = dict(good, terrain=[9] + good["terrain"][1:])
    rc, msg = _create(bad)
    assert rc == -1 and "terrain[0]" in msg
    unwalled = dict(good, terrain=[0] + good["terrain"][1:])
    rc, msg = _create(unwalled)
    assert rc == -1 and "walled" in msg
    p = native.make_params(good)
    p.num_players = 5
    rc, msg = _create(p)
    assert rc == -2 and "1..4 players" in msg
    slow = dict(good, recipe_times=[200] * 16)
    rc, msg = _create(slow)
    assert rc == -2 and "recipe_times" in msg
    start_in_wall = dict(good, start_player_x=[0, 3])
    rc, msg = _create(start_in_wall)
    assert rc == -1 and "start position" in msg
    if not torch.cuda.is_available():
        rc, msg = _create(good)
        assert rc == -4 and "no CPU fallback" in msg            # never a silent CPU path
    assert L.oc_destroy(None) == 0
    assert L.oc_step(None, None, 0, None, None, None, None, None) == -1


def test_python_wrapper_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diverse_conventions_amd import OvercookedMadrona
    with pytest.raises(native.OvercookedNativeError) as e:
        OvercookedMadrona("simple", 4, 0)
    assert e.value.status == -4


CUSTOM = [os.path.join(ROOT, "tests", "golden", "layouts", n + ".layout") for n in ("six_pots", "tiny_mixed", "wide_hall")]


@pytest.mark.parametrize("name", [n for n in layout_names() if get_base_layout_params(n, 200)["num_players"] == 2] + CUSTOM,
                         ids=lambda n: os.path.basename(n).replace(".layout", ""))
def test_per_engine_kernel_build_compiles_for_gfx950(name):
    size = native.jit_compile_check(get_base_layout_params(name, 200), arch="gfx950")
    assert size > 10_000


def test_plain_c_client_links_and_gets_error_codes(tmp_path):
    """The boundary is a C ABI: a C99 translation unit includes the header, links the library and gets
    status codes + messages (no exceptions, no torch types)."""
    import subprocess
    src = tmp_path / "client.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "overcooked_hip.h"
#include "overcooked_ops.h"
#include "overcooked_experiments.h"
#include "overcooked_debug.h"
int main(void) {
    oc_layout_params p; oc_engine *e = NULL; int rc;
    memset(&p, 0, sizeof p);
    rc = oc_create(&p, 4, 0, &e);                       /* zeroed params: rejected, not crashed */
    printf("%d|%s|%d|%lld\n", rc, oc_last_error(), oc_abi_version(), (long long)oc_num_kitchens(NULL));
    return rc == OC_ERR_UNSUPPORTED ? 0 : 1;
}''')
    exe = tmp_path / "client"
    libdir = os.path.dirname(native.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-loc_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().split("|")
    assert out[0] == "-2" and "players" in out[1] and out[2] == "7" and out[3] == "0"


def test_code_object_cache_checks_what_it_loads(tmp_path):
    """The on-disk hiprtc cache: files carry a header (magic, length, checksum) and are ignored -- and rewritten -- when they do
    not check out; a cache directory that others could have written to is not used at all."""
    import stat
    import subprocess
    import sys
    cache = tmp_path / "cache"
    snippet = ("import sys; sys.path.insert(0, %r); from diverse_conventions_amd import native, get_base_layout_params; "
               "print(native.jit_compile_check(get_base_layout_params(sys.argv[1], 200)))" % ROOT)

    def compile_in_fresh_process(layout):
        env = dict(os.environ, OC_JIT_CACHE=str(cache))
        env.pop("OC_SANITIZE", None)
        return int(subprocess.run([sys.executable, "-c", snippet, layout], check=True, capture_output=True, text=True, env=env).stdout.split()[-1])

    size = compile_in_fresh_process("simple")
    files = sorted(cache.glob("*.hsaco"))
    assert len(files) == 1
    good = files[0].read_bytes()
    assert good[:8] == b"OCJIT002" and int.from_bytes(good[8:16], "little") == len(good) - 24 == size and good[24:28] == b"\x7fELF"
    assert stat.S_IMODE(cache.stat().st_mode) == 0o700
    # a flipped byte in the code object: the checksum refuses it, the compile runs again and rewrites the file
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x40
    files[0].write_bytes(bytes(bad))
    assert compile_in_fresh_process("simple") == size and files[0].read_bytes() == good
    # a truncated file and a file with trailing bytes are refused as well
    files[0].write_bytes(good[:-100])
    assert compile_in_fresh_process("simple") == size and files[0].read_bytes() == good
    files[0].write_bytes(good + b"x")
    assert compile_in_fresh_process("simple") == size and files[0].read_bytes() == good
    # a directory others may write to is not trusted: nothing is read from or written to it
    os.chmod(cache, 0o770)
    compile_in_fresh_process("unident_s")
    assert sorted(cache.glob("*.hsaco")) == files
    os.chmod(cache, 0o700)
    compile_in_fresh_process("unident_s")
    assert len(sorted(cache.glob("*.hsaco"))) == 2
