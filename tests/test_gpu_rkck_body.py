"""The Cash-Karp kernels with one lane per body (kernels_rkck_body.hip, small ensembles) against the
one-lane-per-system kernels (kernels_rkck.hip): both run rkck_common.hpp for everything but the forces, and a
system's normalised error is a maximum over its bodies, so step sizes, accept/reject sequences, times and every
coordinate must be identical bit for bit -- and within the usual bars of the oracle's restatement of
integrators/rkck_integrator.hpp."""
import numpy as np
import pytest

from conftest import perturbed, rel_diff, scaled, to_host

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


@pytest.mark.parametrize("nbod,nsys", [(3, 300), (4, 129), (5, 100), (6, 64), (7, 19), (8, 70), (9, 8), (10, 33)])
def test_adaptive_lane_per_body_is_bit_identical(gpu, oracle, nbod, nsys):
    """rkck_adaptive on systems of very different dynamical times (sizes spread over a factor 10): step sizes shrink
    and grow, attempts are rejected, systems of a wave finish far apart; own start times, mixed states."""
    S, O = gpu, oracle
    e = scaled(O, nsys, nbod, 0.3, 3.0, seed=600 + nbod)
    e.time[:] = np.linspace(0.0, 0.5, nsys)
    e.state[5::13] = O.INACTIVE
    e.state[3::11] = O.DISABLED
    cfg = {"integrator": "rkck_adaptive", "min_time_step": "1e-6", "max_time_step": "0.1", "error_tolerance": "1e-22"}
    res = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 2.0, counters=True)
        res.append((h, integ.step_counts()))
    (h1, c1), (h2, c2) = res
    same_bits(h1, h2)
    assert (c1[0] == c2[0]).all() and (c1[1] == c2[1]).all()
    assert (c2[1] > c2[0]).any()                        # some attempts were rejected
    assert len(np.unique(c2[0])) > 5                    # and the systems took different numbers of steps
    off = np.zeros(nsys, bool)
    off[3::11] = True
    for k in ("pos", "vel", "time"):
        assert (getattr(h2, k)[off] == getattr(e, k)[off]).all(), k
    o = e.clone()
    o.state[o.state == O.INACTIVE] = O.ACTIVE
    acc, att = O.integrate(o, O.params("rkck_adaptive", 2.0, min_time_step=1e-6, max_time_step=0.1,
                                       error_tolerance=1e-22), threads=4, counters=True)
    assert (h2.state == o.state).all()
    assert np.abs(c2[0].astype(np.int64) - acc).max() <= 3
    assert rel_diff(h2.pos[:, 1:], o.pos[:, 1:]) < 1e-9


@pytest.mark.parametrize("nbod,nsys", [(3, 200), (6, 77), (10, 20)])
def test_fixed_lane_per_body_is_bit_identical(gpu, oracle, nbod, nsys):
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=700 + nbod)
    e.time[:] = np.linspace(0.0, 0.3, nsys)
    e.state[3::11] = O.DISABLED
    cfg = {"integrator": "rkck_fixed", "min_time_step": "1e-4", "max_time_step": "4e-3", "error_tolerance": "1e-27"}
    res = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 1.0, counters=True)
        res.append((h, integ.step_counts()))
    (h1, c1), (h2, c2) = res
    same_bits(h1, h2)
    assert (c1[0] == c2[0]).all() and (c1[1] == c2[1]).all()
    o = e.clone()
    acc, att = O.integrate(o, O.params("rkck_fixed", 1.0, min_time_step=1e-4, max_time_step=4e-3,
                                       error_tolerance=1e-27), threads=4, counters=True)
    assert (h2.state == o.state).all() and (h2.time == o.time).all() and (c2[0] == acc.astype(np.uint64)).all()
    assert rel_diff(h2.pos[:, 1:], o.pos[:, 1:]) < 1e-10


@pytest.mark.parametrize("nbod,nsys", [(3, 150), (6, 45)])
def test_lane_per_body_ejection_monitor_is_bit_identical(gpu, oracle, nbod, nsys):
    """stop_on_ejection with deactivate_on_ejection alone (C3's monitor): every planet's lane tests its own planet
    after an accepted step; stop times, flags and coordinates equal the lane-per-system kernel's and the oracle's
    states."""
    S, O = gpu, oracle
    e = O.generate(nsys, nbod, spacing_factor=1.4 ** (1.0 / (nbod - 2)))   # circular orbits at r = 10 ... 14
    e.vel[:, 1:] *= np.linspace(0.9, 2.0, nsys)[:, None, None]
    cfg = {"integrator": "rkck_adaptive", "min_time_step": "1e-5", "max_time_step": "0.25", "error_tolerance": "1e-20",
           "deactivate_on_ejection": "1", "rmax": "21"}
    res = []
    for mode in (1, 2, 0):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 60.0, counters=True)
        res.append((h, integ.step_counts()[0]))
    (h1, c1), (h2, c2), (h0, c0) = res
    assert (h2.state == O.DISABLED).sum() > 0 and len(np.unique(h2.time)) > 5 and (h2.state == O.INACTIVE).sum() > 0
    same_bits(h1, h2)
    same_bits(h1, h0)
    assert (c1 == c2).all()
    o = e.clone()
    O.integrate(o, O.params("rkck_adaptive", 60.0, min_time_step=1e-5, max_time_step=0.25, error_tolerance=1e-20),
                O.monitor(rmax=21.0, deactivate_on_ejection=1), threads=4)
    assert (h2.state == o.state).all()
    # a logging or printing monitor is not served by the lane-per-body kernel: lane_mode=2 falls back, same result
    h = to_host(S, e)
    run_gpu(S, h, dict(cfg, lane_mode="2", verbose_on_ejection="1"), 60.0)
    assert (h.state == h1.state).all() and (h.pos == h1.pos).all()
