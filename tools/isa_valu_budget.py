#!/usr/bin/env python3
"""Per-phase instruction budget of one 32-row block of the fused policy kernel, read off the compiler's own listing (no GPU needed):

    python tools/isa_valu_budget.py [kernel-symbol-substring] [positions]

compiles csrc/policy_ops.hip to assembly for gfx950 (product flags), takes the listing of one instantiation of oc_policy_kernel
(default: <3 digits, fc1 from L2, conv digits in LDS>, the eight-wavefront build cramped_room runs) and cuts the block body at its
matrix instructions -- the static order is: first conv (18 i8 MFMAs) | position loop body (18 i8 + 12 f16) | peeled last position
(12 f16) | fc2 (24 f16) | head (12 f16) | epilogue up to the block loop's back edge -- counting, per segment, the VALU instructions
by mnemonic, the LDS / vector-memory / scalar instructions and the s_nop wait states.  Dynamic count per block = first + (P - 1) x
loop + last + fc2 + head + epilogue for P conv positions."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing(symbol_part):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "p.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC", "-I",
                        os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "diverse-conventions_amd", "csrc", "policy_ops.hip")], check=True, stderr=subprocess.DEVNULL)
        lines, on = [], False
        for ln in open(out):
            if not on and re.match(r"^_ZN.*oc_policy_kernel" + symbol_part + r".*:", ln):
                on = True
            if on:
                lines.append(ln.rstrip("\n"))
                if ln.startswith(".Lfunc_end"):
                    break
        return lines


def main():
    part = sys.argv[1] if len(sys.argv) > 1 else "ILi3ELb0ELb1ELb0ELb1ELb0ELb0E"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ins = []
    for ln in listing(part):
        m = re.match(r"^\t([a-z_0-9]+)\b(.*)", ln)
        if m and not m.group(1).startswith((".", ";")):
            ins.append((m.group(1), m.group(2)))
    mf = [i for i, (op, _) in enumerate(ins) if op.startswith("v_mfma")]
    kinds = [ins[i][0] for i in mf]
    n_i8 = sum(1 for k in kinds if "i8" in k)
    conv = n_i8 // 2                                   # first conv + loop conv
    dense = [k for k in kinds if "i8" not in k]
    fc1 = (len(dense) - 0) // 5 if len(dense) % 5 == 0 else None       # 12 + 12 + 24 + 12 = 5 x 12
    if fc1 is None or len(kinds) != 2 * conv + 5 * fc1:
        raise SystemExit(f"unexpected MFMA sequence: {collections.Counter(kinds)}")
    cuts = [mf[0] - 40, mf[conv - 1] + 1, mf[2 * conv + fc1 - 1] + 1, mf[2 * conv + 2 * fc1 - 1] + 1, mf[2 * conv + 4 * fc1 - 1] + 1, mf[-1] + 1]
    # epilogue end: the next backward branch after the last MFMA whose target label precedes the first MFMA: approximated by the next s_cbranch / s_branch run of 400 instructions
    end = min(len(ins), mf[-1] + 1 + 700)
    for i in range(mf[-1] + 1, end):
        if ins[i][0] == "s_endpgm":
            end = i
            break
    names = ["first conv (+ start of the block)", "position loop body", "peeled last position", "fc2", "head", "epilogue (to the end of the block body)"]
    bounds = list(zip([max(cuts[0], 0)] + cuts[1:], cuts[1:] + [end]))
    total = collections.Counter()
    weights = [1, P - 1, 1, 1, 1, 1]
    print(f"kernel *{part}*: {len(kinds)} static MFMAs ({conv} i8 per conv, {fc1} f16 per fc1 position); dynamic counts for P = {P} positions")
    for name, (a, b), w in zip(names, bounds, weights):
        seg = ins[a:b]
        valu = collections.Counter(op for op, _ in seg if op.startswith("v_") and not op.startswith("v_mfma"))
        other = collections.Counter("mfma" if op.startswith("v_mfma") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_")) else
                                    "s_nop" if op == "s_nop" else "s_waitcnt" if op == "s_waitcnt" else "salu" if op.startswith("s_") else None for op, _ in seg)
        other.pop(None, None)
        nops = sum(int(arg.strip().split()[0]) + 1 for op, arg in seg if op == "s_nop")
        print(f"\n== {name}: {sum(valu.values())} VALU x {w}; {dict(other)}; s_nop wait states {nops}")
        print("   " + ", ".join(f"{k} {v}" for k, v in valu.most_common(14)))
        for k, v in valu.items():
            total[k] += v * w
        total["__mfma"] += other.get("mfma", 0) * w
        total["__nop_states"] += nops * w
    mfma = total.pop("__mfma")
    nops = total.pop("__nop_states")
    print(f"\nper 32-row block: {sum(total.values())} non-MFMA VALU instructions beside {mfma} MFMAs ({mfma * 32} matrix cycles), {nops} s_nop wait states")
    print("   " + ", ".join(f"{k} {v}" for k, v in total.most_common(20)))


if __name__ == "__main__":
    main()
