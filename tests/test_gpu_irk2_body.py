"""The Gauss IRK kernel with one lane per body (kernels_irk2_body.hip) against the one-lane-per-system kernel
(kernels_irk2.hip): both run the per-coordinate arithmetic of irk2_common.hpp and add the convergence norm up in the
same order, so sweeps per step, steps, times and every coordinate must be identical bit for bit -- and, like every
irk2 run, within 1e-10 of the oracle's restatement of integrators/irk2_integrator.hpp / irk2_cpu.hpp."""
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


@pytest.mark.parametrize("name,method", [("irk2", "6"), ("irk2", "4"), ("irk2", "2"), ("irk2_cpu", "6")])
@pytest.mark.parametrize("nbod,nsys", [(3, 300), (4, 129), (5, 100), (6, 64), (8, 70), (10, 33)])
def test_lane_per_body_kernel_is_bit_identical(gpu, oracle, nbod, nsys, name, method):
    """lane_mode=2 against lane_mode=1: ragged wave tails, own start times (systems of a wave take different numbers
    of steps and of sweeps), mixed states, a clipped last step, an iteration budget that splits the run."""
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=500 + nbod)
    e.time[:] = np.linspace(0.0, 0.37, nsys)
    e.state[5::13] = O.INACTIVE                         # re-activated by integrate()
    e.state[3::11] = O.DISABLED
    dt = {"6": 0.05, "4": 0.02, "2": 0.004}[method]
    cfg = {"integrator": name, "time_step": str(dt), "method": method}
    T = 1.237
    res = {}
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), T, counters=True)
        res[mode] = (h, integ.step_counts())
    (h1, c1), (h2, c2) = res[1], res[2]
    same_bits(h1, h2)
    assert (c1[0] == c2[0]).all() and c1[0].max() == int(np.ceil(T / dt))
    off = np.zeros(nsys, bool)
    off[3::11] = True
    for k in ("pos", "vel", "time"):                    # never ACTIVE: not a bit may change
        assert (getattr(h2, k)[off] == getattr(e, k)[off]).all(), k
    # an iteration budget splits the run; a launch starts from the Taylor guess again (L137-138), in both kernels
    split = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = S.Integrator.create(S.config(dict(cfg, lane_mode=str(mode), max_iterations="7")))
        integ.set_ensemble(h)
        integ.set_destination_time(T)
        assert integ.integrate() > 2
        split.append(h)
    same_bits(split[0], split[1])
    o = e.clone()
    o.state[o.state == O.INACTIVE] = O.ACTIVE
    O.integrate(o, O.params(name, T, time_step=dt, method=int(method)), threads=4)
    assert (h2.state == o.state).all()
    np.testing.assert_allclose(h2.time, o.time, rtol=0, atol=1e-12)
    assert rel_diff(h2.pos[:, 1:], o.pos[:, 1:]) < 1e-10 and rel_diff(h2.vel[:, 1:], o.vel[:, 1:]) < 1e-10


def test_lane_mode_selection(gpu, oracle):
    """lane_mode=0 picks the lane-per-body kernel from nbod 4 on and for small ensembles; either way the same bits."""
    S, O = gpu, oracle
    for nbod, nsys in ((3, 200), (5, 90)):
        e = perturbed(O, nsys, nbod, seed=9)
        out = []
        for mode in ("0", "1", "2"):
            h = to_host(S, e)
            run_gpu(S, h, {"integrator": "irk2", "time_step": "0.02", "lane_mode": mode}, 0.5)
            out.append(h)
        same_bits(out[0], out[1])
        same_bits(out[0], out[2])


@pytest.mark.parametrize("nbod,nsys", [(3, 150), (6, 45)])
def test_lane_per_body_ejection_monitor_is_bit_identical(gpu, oracle, nbod, nsys):
    """irk2_cpu binds stop_on_ejection (plugins/irk2_cpu.cpp:36); with deactivate_on_ejection alone every planet's
    lane tests its own planet.  Stop steps, flags and coordinates equal the lane-per-system kernel's and the oracle's."""
    S, O = gpu, oracle
    e = O.generate(nsys, nbod, spacing_factor=1.4 ** (1.0 / (nbod - 2)))   # circular orbits at r = 10 ... 14
    e.vel[:, 1:] *= np.linspace(0.9, 2.0, nsys)[:, None, None]
    cfg = {"integrator": "irk2_cpu", "time_step": "0.25", "deactivate_on_ejection": "1", "rmax": "21"}
    res = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 60.0, counters=True)
        res.append((h, integ.step_counts()[0]))
    (h1, c1), (h2, c2) = res
    assert (h2.state == O.DISABLED).sum() > 0 and len(np.unique(h2.time)) > 5 and (h2.state == O.INACTIVE).sum() > 0
    same_bits(h1, h2)
    assert (c1 == c2).all()
    o = e.clone()
    O.integrate(o, O.params("irk2_cpu", 60.0, time_step=0.25), O.monitor(rmax=21.0, deactivate_on_ejection=1), threads=4)
    assert (h2.state == o.state).all() and (h2.time == o.time).all()


def test_no_convergence_in_the_lane_per_body_kernel(gpu, oracle):
    """irk2_integrator.hpp:231-239 through the lane-per-body kernel: 100 sweeps without convergence -> one lprintf
    record per system "no convergence of iteration: 100", the system DISABLED, the step still taken; irk2_cpu
    (irk2_cpu.hpp:246-250) gives up instead and leaves the system as it was.  Same outcome as lane per system."""
    S, O = gpu, oracle
    e = perturbed(O, 70, 3, seed=4)
    e.pos *= 0.05
    e.vel /= np.sqrt(0.05)
    out = {}
    for mode in ("1", "2"):
        h = to_host(S, e)
        integ = S.Integrator.create(S.config({"integrator": "irk2", "time_step": "10", "lane_mode": mode}))
        integ.set_ensemble(h)
        integ.set_destination_time(40.0)
        lm = S.LogManager(1 << 20, "host")
        lm.replay_printf = False
        integ.set_log_manager(lm)
        integ.integrate()
        texts = [r["text"] for r in O.parse_log(bytes(lm.host))]
        assert texts == ["no convergence of iteration: 100\n"] * 70, mode
        out[mode] = h
    same_bits(out["1"], out["2"])
    assert (out["2"].state == O.DISABLED).all() and (out["2"].time == 10.0).all()
    for mode in ("1", "2"):
        h = to_host(S, e)
        integ = S.Integrator.create(S.config({"integrator": "irk2_cpu", "time_step": "10", "max_attempts": "3",
                                              "lane_mode": mode}))
        integ.set_ensemble(h)
        integ.set_destination_time(40.0)
        assert integ.integrate() == 3
        assert (h.state == O.ACTIVE).all() and (h.time == 0.0).all() and (h.pos == e.pos).all()
