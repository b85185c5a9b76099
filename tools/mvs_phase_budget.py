#!/usr/bin/env python3
"""Static per-phase instruction budget of one MVS step at nbod 8: compiles
tools/mvs_phase_budget.hip to gfx950 assembly (no GPU needed) and counts, per phase kernel, the VALU instructions on
the straight-line path (every basic block once; the out-of-line general Stumpff routine is not entered), less those of
the load/store-only kernel.  Usage: python tools/mvs_phase_budget.py [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(asm):
    out, cur = {}, None
    for l in asm.split("\n"):
        m = re.match(r"^(_ZN3swh\w+):", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            t = l.strip()
            if t and not t.startswith((";", ".")):
                out[cur].append(t.split()[0])
    return out


def main():
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "b.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + ROOT + "/include",
                               "-Wno-unused-function", "-DSWH_DISPATCH_ONLY=8", "-S", "--cuda-device-only", "-o", s,
                               os.path.join(ROOT, "tools", "mvs_phase_budget.hip")] + sys.argv[1:],
                              stderr=subprocess.DEVNULL)
        ks = kernels(open(s).read())
    rows = {}
    for name, ops in ks.items():
        m = re.search(r"swh\d+(phase_\w+?)ENS", name)
        if not m:
            continue
        c = collections.Counter(ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        f64 = sum(v for k, v in c.items() if k.startswith("v_") and "f64" in k and not k.startswith("v_cmp"))
        trans = sum(v for k, v in c.items() if k.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")))
        rows[m.group(1)] = (valu, f64, trans, sum(v for k, v in c.items() if k.startswith("s_cbranch")))
    base = rows.pop("phase_none")
    print("phase                      VALU   FP64  rcp/rsq  branches   (less the load/store-only kernel: %d VALU)" % base[0])
    for k, r in sorted(rows.items()):
        print("%-26s %5d  %5d  %5d  %5d" % (k, r[0] - base[0], r[1] - base[1], r[2] - base[2], r[3] - base[3]))


if __name__ == "__main__":
    main()
