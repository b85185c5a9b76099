"""GPU parity of `hermite` + mce_stat (monitors/mercury_mce_stat.hpp, plugins/hermite.cu:40) and of
the lprintf records (gpulog MSG_PRINTF) against the oracle.

PARITY UNPINNED for mce_stat: the reference ships no fixture, no CPU twin and no CUDA build of
this monitor; the oracle is a literal restatement of the header for nbattr >= 3 (the only case
where the reference's attribute accesses stay inside the Body struct)."""
import io

import numpy as np
import pytest

from conftest import rel_diff, to_host

pytestmark = pytest.mark.gpu


def mce_ensemble(O, nsys, nbod=3, planet_mass=1e-3):
    e = O.Ensemble(nsys, nbod, 3, 1)
    g = O.generate(nsys, nbod, planet_mass=planet_mass)
    for k in ("pos", "vel", "mass", "time", "state", "id"):
        setattr(e, k, getattr(g, k).copy())
    e.body_attr[:, 0, 0] = 0.1
    e.body_attr[:, 1:, 0] = 0.05
    return e


def retrograde_pair(O, nsys, nbod=3):
    """planets 1 and 2 on the same circle in opposite senses -> they hit; per-system phase"""
    e = mce_ensemble(O, nsys, nbod)
    r = 10.0
    v = np.sqrt(1 / r)
    dth = np.linspace(0.05, 0.4, nsys)
    z = 0 * dth
    e.pos[:, 1] = np.stack([r + z, z, z], -1)
    e.vel[:, 1] = np.stack([z, v + z, z], -1)
    e.pos[:, 2] = np.stack([r * np.cos(dth), r * np.sin(dth), z], -1)
    e.vel[:, 2] = np.stack([v * np.sin(dth), -v * np.cos(dth), z], -1)
    e.mass[:, 2] = np.where(np.arange(nsys) % 2 == 0, 1e-3, 2e-3)   # odd systems: L312 swaps i and j
    return e


def run(S, O, e, cfg, dest, okw, mon):
    h = to_host(S, e)
    integ = S.Integrator.create(S.config(cfg))
    integ.set_ensemble(h)
    integ.set_destination_time(dest)
    lm = S.LogManager(1 << 20, "host")
    lm.replay_printf = False
    integ.set_log_manager(lm)
    integ.integrate()
    o = e.clone()
    olog = O.Log(1 << 20)
    oname = cfg["integrator"].replace("_close_encounter", "").replace("_log", "")
    O.integrate(o, O.params(oname, dest, **okw), mon, olog)
    return h, lm, o, olog


def texts(O, raw):
    recs = O.parse_log(bytes(raw))
    return sorted(r["text"] for r in recs if r["msgid"] == -2), [r for r in recs if r["msgid"] == 1]


def test_planet_planet_merge_and_the_swapped_loop_halt(gpu, oracle):
    S, O = gpu, oracle
    e = retrograde_pair(O, 32)
    h, lm, o, olog = run(S, O, e, {"integrator": "hermite", "time_step": "0.01"}, 10.0, dict(time_step=0.01),
                         O.monitor(kind=O.MON_MCE_STAT, mce_time_step=0.01))
    assert (h.state == o.state).all()
    assert (o.state[0::2] == O.INACTIVE).all() and (o.state[1::2] == O.DISABLED).all()
    assert (h.mass == o.mass).all()                       # merged masses went through float
    assert (h.time == o.time).all()                       # same halting step
    # the planets pass within ~0.05 of each other before they merge: a 1-ulp change of the input
    # moves the oracle's own output by 4e-9 (pos) / 4e-10 (vel), so 1e-10 is not reachable here
    assert rel_diff(h.pos, o.pos) < 1e-7 and np.abs(h.vel - o.vel).max() < 1e-7
    np.testing.assert_allclose(h.body_attr, o.body_attr, rtol=1e-12)
    np.testing.assert_allclose(h.sys_attr, o.sys_attr, rtol=1e-7)   # energy lost: cancellation of two ~1e-4 terms
    gt, gs = texts(O, lm.host)
    ot, os_ = texts(O, olog.bytes())
    assert len(gt) == 32 + 16 and gt == ot               # "Hitting detected" x32, "3 planets" x16 (identical text)
    assert len(gs) == len(os_) == 16
    key = lambda r: r["sys"]
    for a, b in zip(sorted(gs, key=key), sorted(os_, key=key)):
        assert a["sys"] == b["sys"] and a["time"] == b["time"] and a["state"] == b["state"]
        assert rel_diff(a["bodies"][:, :3], b["bodies"][:, :3]) < 1e-7


def test_sun_hit_merges_planet_into_the_star(gpu, oracle):
    S, O = gpu, oracle
    e = mce_ensemble(O, 16)
    e.pos[:, 1] = [3.0, 0, 0]
    e.vel[:, 1, 0] = -0.3
    e.vel[:, 1, 1] = np.linspace(0.001, 0.03, 16)
    h, lm, o, olog = run(S, O, e, {"integrator": "hermite", "time_step": "0.005"}, 12.0, dict(time_step=0.005),
                         O.monitor(kind=O.MON_MCE_STAT, mce_time_step=0.005))
    assert (o.mass[:, 1] == 0).all() and (h.mass == o.mass).all()
    assert (h.state == o.state).all() and (h.time == o.time).all()
    assert rel_diff(h.pos[:, 2:], o.pos[:, 2:]) < 1e-9
    gt, _ = texts(O, lm.host)
    ot, _ = texts(O, olog.bytes())
    assert gt == ot and len(gt) == 16 and all(t.startswith("Hitting with the sun detected") for t in gt)


def test_quiet_systems_follow_plain_hermite(gpu, oracle):
    """well separated planets: mce_stat never fires, trajectories equal `hermite` without it, and
    init has written rce / rphysics (attributes 1, 2)."""
    S, O = gpu, oracle
    e = mce_ensemble(O, 200, nbod=4, planet_mass=1e-4)
    rng = np.random.default_rng(5)
    e.pos[:, 1:] *= 1 + 1e-3 * rng.standard_normal((200, 3, 3))
    h, lm, o, olog = run(S, O, e, {"integrator": "hermite", "time_step": "0.001", "rceh": "2e-6"}, 2.0,
                         dict(time_step=0.001), O.monitor(kind=O.MON_MCE_STAT, mce_time_step=0.001, rceh=2e-6))
    assert (h.state == o.state).all() and (h.time == o.time).all() and len(lm.host) == 0 and olog.c.len == 0
    assert rel_diff(h.pos[:, 1:], o.pos[:, 1:]) < 1e-10
    np.testing.assert_allclose(h.body_attr, o.body_attr, rtol=1e-12)
    assert (h.body_attr[:, 1:, 2] > 0).all() and (h.body_attr[:, 0, 1:] == 0).all()
    # same input with one body attribute -> `hermite` binds stop_on_ejection; same trajectory
    e1 = O.Ensemble(200, 4)
    for k in ("pos", "vel", "mass", "time", "state", "id"):
        setattr(e1, k, getattr(e, k).copy())
    h1 = to_host(S, e1)
    integ = S.Integrator.create(S.config({"integrator": "hermite", "time_step": "0.001"}))
    integ.set_ensemble(h1)
    integ.set_destination_time(2.0)
    integ.integrate()
    assert rel_diff(h1.pos[:, 1:], h.pos[:, 1:]) < 1e-13 and (h1.time == h.time).all()


def test_mce_refuses_other_integrators_and_missing_attributes(gpu):
    S = gpu
    d = S.DeviceEnsemble(64, 3, 1, 1)
    a = S.capi.LaunchStruct()
    a.integrator, a.monitor, a.time_step, a.destination_time, a.max_iterations = S.capi.HERMITE, S.capi.MON_MCE_STAT, 0.01, 1.0, 10
    assert S.capi.lib.swarm_hip_launch_integrator(d.ref, __import__("ctypes").byref(a)) != 0
    assert b"3 body attributes" in S.capi.lib.swarm_hip_last_error()
    d3 = S.DeviceEnsemble(64, 3, 3, 1)
    a.integrator = S.capi.MVS
    assert S.capi.lib.swarm_hip_launch_integrator(d3.ref, __import__("ctypes").byref(a)) != 0
    assert b"hermite and irk2 only" in S.capi.lib.swarm_hip_last_error()


@pytest.mark.parametrize("name,cfg,okw", [
    ("hermite_adap", {"time_step_factor": "0.017", "min_time_step": "1e-7"}, dict(time_step_factor=0.017, min_time_step=1e-7)),
    ("mvs", {"time_step": "0.01"}, dict(time_step=0.01)),
])
def test_verbose_on_ejection_lprintf_records(gpu, oracle, name, cfg, okw):
    """verbose_on_ejection=1: one "Orbit is hyperbolic/nearly parabolic" line per detection
    (stop_on_ejection.hpp:120-130), same text as the oracle's, replayed to stderr by flush()."""
    S, O = gpu, oracle
    nsys = 64
    e = O.generate(nsys, 3, ejection_factor=2.0)
    e.vel[:, 1:] *= np.linspace(0.6, 1.0, nsys)[:, None, None]
    cfg = dict(cfg, integrator=name, rmax="14", deactivate_on_ejection="1", verbose_on_ejection="1", log_on_ejection="1")
    h, lm, o, olog = run(S, O, e, cfg, 30.0, okw,
                         O.monitor(rmax=14.0, deactivate_on_ejection=1, verbose_on_ejection=1, log_on_ejection=1))
    assert (h.state == o.state).all() and (o.state == O.DISABLED).sum() > 10
    gt, gs = texts(O, lm.host)
    ot, os_ = texts(O, olog.bytes())
    assert len(gt) == len(ot) == (o.state == O.DISABLED).sum() and len(gs) == len(os_)
    strip = lambda t: t.split(" T=")[0]                    # "_sys=%d, bod=%d" part is exact
    assert sorted(map(strip, gt)) == sorted(map(strip, ot))
    if name != "mvs":
        assert gt == ot                                    # %lg prints 6 digits: identical text
    out = io.StringIO()
    assert S.gpulog.replay_printf(out, bytes(lm.host)) == len(gt)
    assert sorted(out.getvalue().splitlines()) == [t.rstrip("\n") for t in gt]


def test_verbose_on_close_encounter_lprintf_records(gpu, oracle):
    S, O = gpu, oracle
    from conftest import head_on
    e = head_on(O, 48)
    cfg = {"integrator": "hermite_adap_close_encounter", "time_step_factor": "0.017", "min_time_step": "1e-7",
           "close_approach": "2.0", "deactivate_on_close_encounter": "1", "verbose_on_close_encounter": "1"}
    h, lm, o, olog = run(S, O, e, cfg, 40.0, dict(time_step_factor=0.017, min_time_step=1e-7),
                         O.monitor(kind=O.MON_EJECTION_OR_CLOSE, close_approach=2.0, deactivate_on_close_encounter=1,
                                   verbose_on_close_encounter=1))
    assert (h.state == o.state).all() and (o.state == O.DISABLED).sum() > 10
    gt, _ = texts(O, lm.host)
    ot, _ = texts(O, olog.bytes())
    assert gt == ot and len(gt) == (o.state == O.DISABLED).sum()
    assert all(t.startswith("Close apporach detected: sys=") for t in gt)


def test_irk2_binds_mce_stat_too(gpu, oracle):
    """plugins/irk2.cu:39: irk2< mce_stat<L>, GravitationAcc >.  mce_stat only looks at a pair when the bounding
    boxes of the two paths over ONE step overlap (boxes grown by 1.2 rce, rce = Hill radius * rceh); with the
    default rceh=1e-6 that is the step in which the point masses pass through each other, where irk2's fixed-point
    iteration cannot converge.  rceh=1 and planet radii of 0.5 make the hit register at a separation of 0.5,
    a well-conditioned merge: states, masses, halting steps and every lprintf line equal the oracle's."""
    S, O = gpu, oracle
    e = retrograde_pair(O, 32)
    e.body_attr[:, 1:, 0] = 0.5
    h, lm, o, olog = run(S, O, e, {"integrator": "irk2", "time_step": "0.01", "rceh": "1.0"}, 10.0, dict(time_step=0.01),
                         O.monitor(kind=O.MON_MCE_STAT, mce_time_step=0.01, rceh=1.0))
    assert (h.state == o.state).all() and (h.mass == o.mass).all() and (h.time == o.time).all()
    assert (o.state[0::2] == O.INACTIVE).all() and (o.state[1::2] == O.DISABLED).all()
    assert rel_diff(h.pos, o.pos) < 1e-9 and np.abs(h.vel - o.vel).max() < 1e-10
    np.testing.assert_allclose(h.body_attr, o.body_attr, rtol=1e-12)
    gt, gs = texts(O, lm.host)
    ot, os_ = texts(O, olog.bytes())
    assert gt == ot and len(gt) == 48 and len(gs) == len(os_) == 16
