"""Energy conservation of the fixed-step Hermite kernel at nbod 9 with and without the velocity fold of hermite_step
(device_common.hpp, nbod >= 9): run under SWARM_HIP_LIB=<variant built with HERMITE_FOLD_FROM = 99 in device_common.hpp> and without.
Prints max and median |dE/E| over the ensemble after 1e5 steps of 1e-3."""
import sys
import numpy as np
sys.path.insert(0, ".")
import swarm_amd as S
from oracle import oracle as O
sys.path.insert(0, "tests")
from conftest import perturbed, to_host

O.build()
S.capi.check(S.capi.lib.swarm_hip_init(0))
nsys, nbod = 4096, 9
e = perturbed(O, nsys, nbod, seed=2)
e0 = O.total_energy(e)
h = to_host(S, e)
integ = S.Integrator.create(S.config({"integrator": "hermite", "time_step": "0.001", "lane_mode": "1"}))
integ.set_ensemble(h)
integ.set_destination_time(100.0)
integ.integrate()
o = e.clone()
o.pos[:], o.vel[:] = h.pos, h.vel
err = np.abs((O.total_energy(o) - e0) / e0)
print("max %.3e median %.3e" % (err.max(), np.median(err)))
