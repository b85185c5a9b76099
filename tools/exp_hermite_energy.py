#!/usr/bin/env python3
"""Energy conservation of the fixed-step Hermite kernel under the loaded library (SWARM_HIP_LIB selects a variant):
py/tests/trivial.py's BasicIntegration case (generated 16 x 3, time_step 1e-4, T = 10: the reference's bar is
max |dE/E| < 1e-13) and a perturbed ensemble of 4096 systems (median / 90 % / max), plus the distance of the final
positions and velocities from the oracle's hermite_cpu on a sample.  Used to judge changes of the corrector's form."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import swarm_amd as S
from oracle import oracle as O


def run(host, T, dt):
    # lane_mode=1: the lane-per-system kernel whatever the ensemble size (both layouts give the same bits in the shipped library)
    integ = S.Integrator.create(S.config({"integrator": "hermite", "time_step": repr(dt), "lane_mode": "1"}))
    integ.set_ensemble(host)
    integ.set_destination_time(T)
    e0 = host.calc_total_energy()
    integ.integrate()
    return np.abs((host.calc_total_energy() - e0) / e0)


NBODS = ((3, 0.1), (6, 0.1), (3, 1.0))


def main():
    global NBODS
    if len(sys.argv) > 1 and sys.argv[1] == "--nbod3":
        NBODS = ((3, 0.1), (3, 1.0))
    S.capi.check(S.capi.lib.swarm_hip_init(0))
    out = {}
    h = S.generate_ensemble(S.config({"nsys": "16", "nbod": "3"}))
    out["basic_integration_max_dE"] = float(run(h, 10.0, 1e-4).max())
    for nbod, s in NBODS:
        e = O.generate(4096, nbod)
        rng = np.random.default_rng(11)
        e.pos[:, 1:, :] *= 1.0 + 1e-2 * rng.standard_normal((4096, nbod - 1, 3))
        e.vel[:, 1:, :] *= 1.0 + 1e-2 * rng.standard_normal((4096, nbod - 1, 3))
        # s = 0.1: shrunk 10x, orbital periods of ~1 .. 3, so that 1e5 steps are many orbits; s = 1: the generated
        # sizes (r = 10, 14: a twentieth of an orbit, the regime of trivial.py's BasicIntegration)
        e.pos *= s
        e.vel /= np.sqrt(s)
        h = S.HostEnsemble(4096, nbod)
        for k in ("pos", "vel", "mass", "time", "state", "id"):
            getattr(h, k)[...] = getattr(e, k)
        de = run(h, 10.0, 1e-4)
        tag = "perturbed_4096x%d%s" % (nbod, "" if s < 1 else "_unshrunk")
        out[tag + "_dE_median_p90_max"] = [float(np.median(de)), float(np.quantile(de, 0.9)), float(de.max())]
        idx = np.arange(0, 4096, 64)
        o = O.generate(len(idx), nbod)
        for k in ("pos", "vel", "mass", "time", "state", "id"):
            getattr(o, k)[...] = getattr(e, k)[idx]
        O.integrate(o, O.params("hermite_cpu", 10.0, time_step=1e-4), threads=8)
        dp = np.sqrt(((h.pos[idx] - o.pos) ** 2).sum(-1)).max()
        dv = np.sqrt(((h.vel[idx] - o.vel) ** 2).sum(-1)).max()
        out[tag + "_vs_hermite_cpu_pos_vel"] = [float(dp), float(dv)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
