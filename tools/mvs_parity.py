#!/usr/bin/env python3
"""MVS against the reference's own ctest bar (cmake/test_integrators.cmake:38: pos/vel 1e-8 vs the
hermite_cpu goldens, test/integrators/MVS.cfg: time_step=0.0003, T=1) -- what the HIP kernel measures.

For nbod 3..8 prints, as JSON lines, the compare_ensembles distances (swarm/utils.cpp:170-214: max over
bodies of the absolute Euclidean difference)
  hip_vs_golden   HIP kernel vs test.N.out.txt
  orc_vs_golden   the CPU restatement of propagators/{mvs,keplerian}.hpp (oracle/) vs test.N.out.txt
  hip_vs_orc      HIP kernel vs that restatement
Test infrastructure: imports oracle/ as the checker.  Run on the GPU box:  python tools/mvs_parity.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import swarm_amd as S
from conftest import GOLDEN, to_host
from oracle import oracle as O


def dist(a, b):
    return (float(np.sqrt(((a.pos - b.pos) ** 2).sum(-1)).max()), float(np.sqrt(((a.vel - b.vel) ** 2).sum(-1)).max()))


def main():
    S.capi.check(S.capi.lib.swarm_hip_init(0))
    extra = dict(kv.split("=", 1) for kv in sys.argv[1:])
    for dt in ("0.0003", "0.01"):
        for n in range(3, 9):
            e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % n))
            ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.out.txt" % n))
            h = to_host(S, e)
            cfg = {"integrator": "mvs", "time_step": dt}
            cfg.update(extra)
            integ = S.Integrator.create(S.config(cfg))
            integ.set_ensemble(h)
            integ.set_destination_time(1.0)
            integ.integrate()
            O.integrate(e, O.params("mvs", 1.0, time_step=float(dt)))
            print(json.dumps({"time_step": dt, "nbod": n, "hip_vs_golden": dist(h, ref), "orc_vs_golden": dist(e, ref),
                              "hip_vs_orc": dist(h, e), "times_equal": bool((h.time == e.time).all())}))


if __name__ == "__main__":
    main()
