"""Ragged wavefronts: systems that start at different times, systems that are not ACTIVE at launch, a small
iteration budget per launch (several launches) -- so the lanes of a wave finish at very different iterations.
These are the runs in which the kernels keep a wave whole (a lane with nothing left to do rides along
with h = 0 or a withheld commit until its wave is done); the results must be what the one-system-at-a-time oracle
gives: states, final times and step counts exactly (fixed step) or to one step (adaptive), coordinates to 1e-10, and
systems that were never ACTIVE bit for bit as they were."""
import numpy as np
import pytest

from conftest import perturbed, rel_diff, to_host

pytestmark = pytest.mark.gpu

EJ = {"deactivate_on_ejection": "1", "rmax": "1000"}
CASES = {
    # name: (plugin, cfg, oracle integrator, oracle params, oracle monitor kwargs, fixed step?)
    "hermite_ejection_monitor": ("hermite", dict(EJ, time_step="0.004"), "hermite_cpu", dict(time_step=0.004),
                                 dict(rmax=1000.0, deactivate_on_ejection=1), True),
    "hermite_adap": ("hermite_adap", {"time_step_factor": "0.0005", "min_time_step": "1e-7"}, "hermite_adap",
                     dict(time_step_factor=0.0005, min_time_step=1e-7), {}, False),
    "rkck_fixed": ("rkck_fixed", {"min_time_step": "1e-4", "max_time_step": "4e-3", "error_tolerance": "1e-27"},
                   "rkck_fixed", dict(min_time_step=1e-4, max_time_step=4e-3, error_tolerance=1e-27), {}, True),
    "rkck_adaptive": ("rkck_adaptive", dict(EJ, min_time_step="1e-5", max_time_step="0.01", error_tolerance="1e-22"),
                      "rkck_adaptive", dict(min_time_step=1e-5, max_time_step=0.01, error_tolerance=1e-22),
                      dict(rmax=1000.0, deactivate_on_ejection=1), False),
    "mvs_ejection_monitor": ("mvs", dict(EJ, time_step="0.01"), "mvs", dict(time_step=0.01),
                             dict(rmax=1000.0, deactivate_on_ejection=1), True),
    "euler_ejection_monitor": ("euler", dict(EJ, time_step="0.002"), "euler", dict(time_step=0.002),
                               dict(rmax=1000.0, deactivate_on_ejection=1), True),
    "midpoint_ejection_monitor": ("midpoint", dict(EJ, time_step="0.004"), "midpoint", dict(time_step=0.004),
                                  dict(rmax=1000.0, deactivate_on_ejection=1), True),
    "hermite_prop_ejection_monitor": ("hermite_prop", dict(EJ, time_step="0.004"), "hermite_prop", dict(time_step=0.004),
                                      dict(rmax=1000.0, deactivate_on_ejection=1), True),
    "verlet": ("verlet", {"max_timestep": "2e-3", "timestep_scale": "1.0"}, "verlet",
               dict(max_timestep=2e-3, timestep_scale=1.0), {}, False),
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("nbod", [3, 5])
def test_ragged_waves_match_the_oracle(gpu, oracle, case, nbod):
    S, O = gpu, oracle
    plugin, cfg, oname, okw, omon, fixed = CASES[case]
    nsys, T = 64 * 5 + 23, 1.0
    rng = np.random.default_rng(nbod * 7 + len(case))
    e = perturbed(O, nsys, nbod, seed=31 + nbod)
    # start times all over [0, 0.9 T): most waves hold lanes that need anything from a few steps to all of them;
    # the last two wave chunks are sparse: 3 and 1 ACTIVE systems
    e.time[:] = np.round(rng.uniform(0.0, 0.9 * T, nsys), 3)
    e.state[rng.choice(nsys, 40, replace=False)] = O.DISABLED
    e.state[rng.choice(nsys, 20, replace=False)] = O.NEEDS_EXAM
    e.state[64 * 4:] = O.DISABLED
    e.state[[64 * 4 + 5, 64 * 4 + 40, 64 * 4 + 63, 64 * 5 + 11]] = O.ACTIVE
    never = e.state != O.ACTIVE
    h = to_host(S, e)
    integ = S.Integrator.create(S.config(dict(cfg, integrator=plugin, max_iterations="37")))
    integ.set_ensemble(h)
    integ.set_destination_time(T)
    integ.enable_step_counters()
    launches = integ.integrate()
    o = e.clone()
    acc, att = O.integrate(o, O.params(oname, T, max_iterations=37, **okw), O.monitor(**omon), threads=4, counters=True)
    gacc, gatt = integ.step_counts()
    assert launches > 1 or plugin.startswith("rkck")
    assert (h.state == o.state).all() and (h.id == e.id).all()
    for k in ("pos", "vel", "time", "mass"):                      # never ACTIVE: not a bit may change
        assert (getattr(h, k)[never] == getattr(e, k)[never]).all(), k
    assert (gacc[never] == 0).all()
    if fixed:
        assert (h.time == o.time).all()
        assert (gacc == acc.astype(np.uint64)).all() and (gatt == att.astype(np.uint64)).all()
    else:
        np.testing.assert_allclose(h.time, o.time, rtol=1e-12)
        assert np.abs(gacc.astype(np.int64) - acc).max() <= 1
    live = ~never
    assert rel_diff(h.pos[live][:, 1:], o.pos[live][:, 1:]) < 1e-10
    assert rel_diff(h.vel[live][:, 1:], o.vel[live][:, 1:]) < 1e-10
