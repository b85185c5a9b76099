"""The MVS kernel with one lane per body (kernels_mvs_body.hip, small ensembles) against the one-lane-per-system kernel
(kernels_generic.hip): both are written over the same per-planet solver (mvs_common.hpp) with the sums in the same
order, so the ensembles they leave must be identical bit for bit -- and, like every MVS run, within 1e-10 of the
oracle's restatement of propagators/mvs.hpp."""
import numpy as np
import pytest

from conftest import perturbed, rel_diff, to_host

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "vel", "time", "state", "id", "mass")


def run_gpu(S, host, cfg, dest, counters=False):
    integ = S.Integrator.create(S.config(cfg))
    integ.set_ensemble(host)
    integ.set_destination_time(dest)
    if counters:
        integ.enable_step_counters()
    integ.integrate()
    return integ


def same_bits(a, b):
    for k in FIELDS:
        assert (getattr(a, k) == getattr(b, k)).all(), k


@pytest.mark.parametrize("compensated", ["0", "1"])
@pytest.mark.parametrize("nbod,nsys", [(3, 1000), (4, 257), (5, 100), (6, 64), (7, 19), (8, 300), (9, 8), (10, 33)])
def test_lane_per_body_kernel_is_bit_identical(gpu, oracle, nbod, nsys, compensated):
    """lane_mode=2 against lane_mode=1 for ragged wave tails (64 % nbod idle lanes, nsys not a multiple of the systems
    per wavefront), systems with their own start times and states, a clipped last step, and an iteration budget that
    splits the run into launches (each launch converts to the propagator's coordinates and back)."""
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=300 + nbod)
    e.time[:] = np.linspace(0.0, 0.037, nsys)           # own times: systems of one wave finish at different steps
    e.state[5::13] = O.INACTIVE                         # re-activated by integrate()
    e.state[3::11] = O.DISABLED
    cfg = {"integrator": "mvs", "time_step": "0.001", "kepler_compensated": compensated}
    res = {}
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 0.25037, counters=True)
        res[mode] = (h, integ.step_counts())
    (h1, c1), (h2, c2) = res[1], res[2]
    same_bits(h1, h2)
    assert (c1[0] == c2[0]).all() and (c1[1] == c2[1]).all() and c1[0].max() == 251
    off = np.zeros(nsys, bool)
    off[3::11] = True
    for k in ("pos", "vel", "time"):                    # never ACTIVE: not a bit may change
        assert (getattr(h2, k)[off] == getattr(e, k)[off]).all(), k
    assert (h2.state[off] == O.DISABLED).all() and (c2[0][off] == 0).all()
    split = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = S.Integrator.create(S.config(dict(cfg, lane_mode=str(mode), max_iterations="100")))
        integ.set_ensemble(h)
        integ.set_destination_time(0.25037)
        assert integ.integrate() == 3
        split.append(h)
    same_bits(split[0], split[1])
    o = e.clone()
    o.state[o.state == O.INACTIVE] = O.ACTIVE
    O.integrate(o, O.params("mvs", 0.25037, time_step=0.001, kepler_compensated=int(compensated)), threads=4)
    assert (h2.state == o.state).all() and (h2.time == o.time).all()
    assert rel_diff(h2.pos[:, 1:], o.pos[:, 1:]) < 1e-10 and rel_diff(h2.vel[:, 1:], o.vel[:, 1:]) < 1e-10


@pytest.mark.parametrize("nbod,nsys", [(3, 200), (5, 77), (8, 40)])
def test_lane_per_body_ejection_monitor_is_bit_identical(gpu, oracle, nbod, nsys):
    """stop_on_ejection with deactivate_on_ejection=1: every planet's lane tests its own planet in the propagator's
    coordinates.  Orbits from bound to fast escapers; the step at which each system is disabled, the flags and every
    coordinate equal the lane-per-system kernel's, and the stop steps are the oracle's."""
    S, O = gpu, oracle
    e = O.generate(nsys, nbod, spacing_factor=1.4 ** (1.0 / (nbod - 2)))   # circular orbits at r = 10 ... 14
    e.vel[:, 1:] *= np.linspace(0.9, 2.0, nsys)[:, None, None]   # ... sped up: beyond sqrt(2) they leave, sooner or later
    rmax = 21.0
    cfg = {"integrator": "mvs", "time_step": "0.02", "deactivate_on_ejection": "1", "rmax": str(rmax)}
    res = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 60.0, counters=True)
        res.append((h, integ.step_counts()[0]))
    (h1, c1), (h2, c2) = res
    assert (h2.state == O.DISABLED).sum() > 0 and len(np.unique(h2.time)) > 5 and (h2.state == O.INACTIVE).sum() > 0   # stopped at many different steps
    same_bits(h1, h2)
    assert (c1 == c2).all()
    o = e.clone()
    O.integrate(o, O.params("mvs", 60.0, time_step=0.02), O.monitor(rmax=rmax, deactivate_on_ejection=1), threads=4)
    assert (h2.state == o.state).all() and (h2.time == o.time).all()
    # a logging or printing monitor is not served by the lane-per-body kernel: lane_mode=2 falls back, same result
    h = to_host(S, e)
    run_gpu(S, h, dict(cfg, lane_mode="2", verbose_on_ejection="1"), 60.0)
    assert (h.state == h1.state).all() and (h.pos == h1.pos).all()


def test_general_stumpff_branch_is_bit_identical(gpu, oracle):
    """Steps of a good fraction of an orbit (|alpha x^2| >= 1e-4: the trigonometric and hyperbolic branches of
    SC_prussing, keplerian.hpp:67-86) mixed in one wave with systems on the series branch: what a planet gets must not
    depend on its neighbours in the wavefront, in either kernel."""
    S, O = gpu, oracle
    nsys, nbod = 150, 4
    e = perturbed(O, nsys, nbod, seed=77)
    e.pos[::3, 1:] *= 0.05                              # tight orbits: many radians per step
    e.vel[::3, 1:] *= np.sqrt(20.0)
    e.vel[1::7, 1:] *= 1.6                              # unbound: alpha < 0
    cfg = {"integrator": "mvs", "time_step": "0.01"}
    out = []
    for mode in ("1", "2", "0"):
        h = to_host(S, e)
        run_gpu(S, h, dict(cfg, lane_mode=mode), 0.5)
        out.append(h)
    same_bits(out[0], out[1])
    same_bits(out[0], out[2])
    # the same systems alone in their waves (the wave-uniform series shortcut then decides differently)
    pick = np.arange(nsys)[1::3]
    sub = O.Ensemble(len(pick), nbod)
    for k in FIELDS:
        getattr(sub, k)[...] = getattr(e, k)[pick]
    for mode in ("1", "2"):
        h = to_host(S, sub)
        run_gpu(S, h, dict(cfg, lane_mode=mode), 0.5)
        for k in ("pos", "vel", "time", "state"):
            assert (getattr(h, k) == getattr(out[0], k)[pick]).all(), (mode, k)
    o = e.clone()
    O.integrate(o, O.params("mvs", 0.5, time_step=0.01), threads=4)
    ok = np.isfinite(o.pos).all(axis=(1, 2))
    assert ok.sum() > nsys // 2
    assert rel_diff(out[0].pos[ok][:, 1:], o.pos[ok][:, 1:]) < 1e-9
