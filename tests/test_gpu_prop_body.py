"""The tutorial propagators euler, midpoint, verlet and hermite_prop with one lane per body (kernels_prop_body.hip, small
ensembles) against the one-lane-per-system kernel (kernels_generic.hip): both call the same per-coordinate updates
(propagators_common.hpp, hermite_step) on forces summed in the same order, so the ensembles they leave must be
identical bit for bit -- and within 1e-10 of the oracle's restatement of propagators/{euler,midpoint,verlet,
hermite_propagator}.hpp.  Verlet's own time step (a sum over all ordered pairs of a system, gathered through LDS in
the lane-per-body kernel) makes its systems take different numbers of steps."""
import numpy as np
import pytest

from conftest import perturbed, rel_diff, to_host

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "vel", "time", "state", "id", "mass")
PLUGINS = ("euler", "midpoint", "hermite_prop", "verlet")
STEP = {"verlet": {"max_timestep": "0.001", "timestep_scale": "1e6"}}      # r ~ 10: the pair terms are ~1e-12, steps well below the maximum
OSTEP = {"verlet": dict(max_timestep=0.001, timestep_scale=1e6)}


def run_gpu(S, host, cfg, dest, counters=False):
    integ = S.Integrator.create(S.config(cfg))
    integ.set_ensemble(host)
    integ.set_destination_time(dest)
    if counters:
        integ.enable_step_counters()
    integ.integrate()
    return integ


def split_records(raw):
    out, off = [], 0
    while off < len(raw):
        ln = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off + 4)[0])
        out.append(raw[off:off + ln])
        off += ln
    return sorted(out)


def same_bits(a, b):
    for k in FIELDS:
        assert (getattr(a, k) == getattr(b, k)).all(), k


@pytest.mark.parametrize("plugin", PLUGINS)
@pytest.mark.parametrize("nbod,nsys", [(3, 1000), (4, 257), (5, 100), (6, 64), (7, 19), (8, 300), (9, 8), (10, 33)])
def test_lane_per_body_kernel_is_bit_identical(gpu, oracle, plugin, nbod, nsys):
    """lane_mode=2 against lane_mode=1 for ragged wave tails, systems with their own start times and states, a clipped
    last step, and an iteration budget that splits the run into launches."""
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=500 + nbod)
    e.time[:] = np.linspace(0.0, 0.037, nsys)           # own times: systems of one wave finish at different steps
    e.state[5::13] = O.INACTIVE                         # re-activated by integrate()
    e.state[3::11] = O.DISABLED
    cfg = dict({"integrator": plugin}, **STEP.get(plugin, {"time_step": "0.001"}))
    res = {}
    for mode in (1, 2):
        h = to_host(S, e)
        integ = run_gpu(S, h, dict(cfg, lane_mode=str(mode)), 0.25037, counters=True)
        res[mode] = (h, integ.step_counts())
    (h1, c1), (h2, c2) = res[1], res[2]
    same_bits(h1, h2)
    assert (c1[0] == c2[0]).all() and (c1[1] == c2[1]).all()
    if plugin == "verlet":
        assert c1[0].max() > 251 and len(np.unique(c1[0])) > min(nsys, 16) // 2      # own steps, never clipped
    else:
        assert c1[0].max() == 251
    off = np.zeros(nsys, bool)
    off[3::11] = True
    for k in ("pos", "vel", "time"):                    # never ACTIVE: not a bit may change
        assert (getattr(h2, k)[off] == getattr(e, k)[off]).all(), k
    assert (h2.state[off] == O.DISABLED).all() and (c2[0][off] == 0).all()
    split = []
    for mode in (1, 2, 0):                              # 0: the library's own choice (the body kernel from nbod 4 on)
        h = to_host(S, e)
        integ = S.Integrator.create(S.config(dict(cfg, lane_mode=str(mode), max_iterations="100")))
        integ.set_ensemble(h)
        integ.set_destination_time(0.25037)
        assert integ.integrate() >= 3
        split.append(h)
    same_bits(split[0], split[1])
    same_bits(split[0], split[2])
    o = e.clone()
    o.state[o.state == O.INACTIVE] = O.ACTIVE
    O.integrate(o, O.params(plugin, 0.25037, **OSTEP.get(plugin, dict(time_step=0.001))), threads=4)
    assert (h2.state == o.state).all()
    if plugin == "verlet":                               # its times are sums of computed steps
        assert np.allclose(h2.time, o.time, rtol=0, atol=1e-12)
    else:
        assert (h2.time == o.time).all()
    assert rel_diff(h2.pos, o.pos) < 1e-10 and rel_diff(h2.vel, o.vel) < 1e-10


@pytest.mark.parametrize("plugin", PLUGINS)
@pytest.mark.parametrize("nbod,nsys", [(3, 200), (5, 77), (8, 40)])
def test_lane_per_body_ejection_monitor_is_bit_identical(gpu, oracle, plugin, nbod, nsys):
    """stop_on_ejection with deactivate_on_ejection=1 and log_on_ejection=1: orbits from bound to fast escapers; the step
    at which each system is disabled, the flags, every coordinate and the records in the log equal the lane-per-system
    kernel's, and the stop steps are the oracle's."""
    S, O = gpu, oracle
    e = O.generate(nsys, nbod, spacing_factor=1.4 ** (1.0 / (nbod - 2)))   # circular orbits at r = 10 ... 14
    e.vel[:, 1:] *= np.linspace(0.9, 2.0, nsys)[:, None, None]   # ... sped up: beyond sqrt(2) they leave, sooner or later
    rmax = 21.0
    step = {"max_timestep": "0.02", "timestep_scale": "1e6"} if plugin == "verlet" else {"time_step": "0.02"}
    cfg = dict({"integrator": plugin, "deactivate_on_ejection": "1", "log_on_ejection": "1", "rmax": str(rmax),
                "log_writer": "null"}, **step)
    res = []
    for mode in (1, 2):
        h = to_host(S, e)
        integ = S.Integrator.create(S.config(dict(cfg, lane_mode=str(mode))))
        integ.set_ensemble(h)
        integ.set_destination_time(60.0)
        integ.enable_step_counters()
        lm = S.LogManager(8 << 20, "host")
        lm.replay_printf = False
        integ.set_log_manager(lm)
        integ.integrate()
        res.append((h, integ.step_counts()[0], split_records(bytes(lm.host))))
    (h1, c1, r1), (h2, c2, r2) = res
    assert (h2.state == O.DISABLED).sum() > 0 and len(np.unique(h2.time)) > 5 and (h2.state == O.INACTIVE).sum() > 0
    same_bits(h1, h2)
    assert (c1 == c2).all()
    assert len(r2) == (h2.state == O.DISABLED).sum() and r1 == r2
    o = e.clone()
    ostep = dict(max_timestep=0.02, timestep_scale=1e6) if plugin == "verlet" else dict(time_step=0.02)
    O.integrate(o, O.params(plugin, 60.0, **ostep), O.monitor(rmax=rmax, deactivate_on_ejection=1), threads=4)
    assert (h2.state == o.state).all() and np.allclose(h2.time, o.time, rtol=0, atol=1e-9 if plugin == "verlet" else 0)
