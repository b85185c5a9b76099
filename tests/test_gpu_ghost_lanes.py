"""Ghost lanes (device_common.hpp take_ghosts): a fixed-step kernel with nothing to monitor turns idle lanes into copies of
one of its active systems when 8 or fewer systems of a wavefront are ACTIVE, so that its FP64 instructions issue at
full rate.  Ghosts store and count nothing: the ACTIVE systems must come out exactly as from a dense run, the others
untouched, bit for bit."""
import numpy as np
import pytest

from conftest import perturbed, to_host

pytestmark = pytest.mark.gpu

CASES = {
    "hermite": {"time_step": "0.001"},
    "mvs": {"time_step": "0.001"},
    "euler": {"time_step": "0.001"},
    "midpoint": {"time_step": "0.001"},
    "hermite_prop": {"time_step": "0.001"},
    "irk2": {"time_step": "0.01"},
}


def run(S, host, cfg, dest):
    integ = S.Integrator.create(S.config(cfg))
    integ.set_ensemble(host)
    integ.set_destination_time(dest)
    integ.enable_step_counters()
    integ.integrate()
    return integ.step_counts()[0]


@pytest.mark.parametrize("plugin", sorted(CASES))
@pytest.mark.parametrize("nbod,nsys,keep", [(3, 1000, 16), (6, 333, 9), (4, 130, 64), (8, 70, 8)])
def test_sparse_ensemble_equals_dense_run(gpu, oracle, plugin, nbod, nsys, keep):
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=900 + nbod)
    e.time[:] = np.linspace(0.0, 0.02, nsys)            # ghosts copy the active system of their wave with the earliest time
    cfg = dict(CASES[plugin], integrator=plugin, lane_mode="1")
    dense = to_host(S, e)
    cd = run(S, dense, cfg, 0.1003)
    sp = e.clone()
    active = np.zeros(nsys, bool)
    active[::keep] = True
    sp.state[~active] = O.DISABLED
    sp.pos[5::7] *= 1e150                               # disabled systems may hold anything; a ghost must not use it
    sp.pos[active] = e.pos[active]
    sparse = to_host(S, sp)
    cs = run(S, sparse, cfg, 0.1003)
    for k in ("pos", "vel", "time", "state"):
        assert (getattr(sparse, k)[active] == getattr(dense, k)[active]).all(), k
        assert (getattr(sparse, k)[~active] == getattr(sp, k)[~active]).all(), k
    assert (cs[active] == cd[active]).all() and (cs[~active] == 0).all()
