"""Snapshot records from the lane-per-body Hermite kernel (kernels_hermite_body.hip): the N lanes of a system write one
EVT_SNAPSHOT record together (log_system_body) -- the star's lane the header, every lane its own body.  Against the
lane-per-system kernel the records must be the same byte for byte (their order in the buffer is free: whichever
wavefront reserves first), and the ensembles bit-identical; against the oracle the record times and systems agree."""
import numpy as np
import pytest

from conftest import perturbed, to_host

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "vel", "time", "state", "id", "mass")


def run_logged(S, e, cfg, dest, cap=8 << 20):
    h = to_host(S, e)
    integ = S.Integrator.create(S.config(cfg))
    integ.set_ensemble(h)
    integ.set_destination_time(dest)
    lm = S.LogManager(cap, "host")
    lm.replay_printf = False
    integ.set_log_manager(lm)
    integ.integrate()
    return h, bytes(lm.host), lm.dropped


def split(raw):
    out, off = [], 0
    while off < len(raw):
        ln = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off + 4)[0])
        out.append(raw[off:off + ln])
        off += ln
    return sorted(out)


@pytest.mark.parametrize("name,cfg", [
    ("hermite_cpu_log", {"time_step": "0.001", "log_interval": "0.0107"}),     # plugins/hermite_cpu.cpp: log_time_interval
    ("hermite_adap_log", {"time_step_factor": "0.017", "min_time_step": "1e-5", "log_interval": "0.5"}),
])
@pytest.mark.parametrize("nbod,nsys", [(3, 300), (5, 77), (8, 40), (10, 13)])
def test_log_time_interval_records_are_identical(gpu, oracle, name, cfg, nbod, nsys):
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=800 + nbod)
    e.time[:] = np.linspace(0.0, 0.05, nsys)            # own times: snapshots fall due on different steps
    e.state[3::11] = O.DISABLED
    e.id[:] = 1000 + np.arange(nsys)
    T = 0.2 if name == "hermite_cpu_log" else 8.0
    out = {}
    for mode in ("1", "2"):
        out[mode] = run_logged(S, e, dict(cfg, integrator=name, log_writer="null", lane_mode=mode, max_iterations="61"), T)
    (h1, raw1, d1), (h2, raw2, d2) = out["1"], out["2"]
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all(), k
    r1, r2 = split(raw1), split(raw2)
    assert len(r2) > nsys and r1 == r2 and d1 == d2 == 0
    recs = O.parse_log(raw2)
    assert all(r["msgid"] == 1 and r["nbod"] == nbod and (r["body_id"] == np.arange(nbod)).all() for r in recs)
    assert {r["sys"] for r in recs} <= set(int(i) for i in e.id[e.state == O.ACTIVE]) | (
        set(int(i) for i in e.id) if name == "hermite_adap_log" else set())
    # the oracle logs the same (system, time) pairs
    o = e.clone()
    olog = O.Log(8 << 20)
    kw = dict(time_step=0.001) if name == "hermite_cpu_log" else dict(time_step_factor=0.017, min_time_step=1e-5)
    O.integrate(o, O.params("hermite_cpu" if name == "hermite_cpu_log" else "hermite_adap", T, max_iterations=61, **kw),
                O.monitor(kind=O.MON_LOG_INTERVAL, log_interval=float(cfg["log_interval"])), olog, threads=1)
    mine = sorted((r["sys"], round(r["time"], 9)) for r in recs)
    ref = sorted((r["sys"], round(r["time"], 9)) for r in olog.records())
    assert mine == ref


EJECTING = {
    "hermite_cpu": {"time_step": "0.02"},
    "rkck_adaptive": {"min_time_step": "1e-5", "max_time_step": "0.25", "error_tolerance": "1e-20"},
    "irk2_cpu": {"time_step": "0.25"},
    "mvs": {"time_step": "0.02"},       # the record is in standard coordinates: conversion there and back (mvs.hpp:159-230)
}


@pytest.mark.parametrize("name", sorted(EJECTING))
@pytest.mark.parametrize("nbod,nsys", [(3, 200), (6, 45)])
def test_log_on_ejection_records_are_identical(gpu, oracle, name, nbod, nsys):
    """stop_on_ejection with log_on_ejection (and deactivate_on_ejection) in the lane-per-body Hermite, Cash-Karp,
    Gauss IRK and MVS kernels: one record per ejected system, taken after the state change (stop_on_ejection.hpp:140-146)."""
    S, O = gpu, oracle
    e = O.generate(nsys, nbod, spacing_factor=1.4 ** (1.0 / (nbod - 2)))   # circular orbits at r = 10 ... 14
    e.vel[:, 1:] *= np.linspace(0.9, 2.0, nsys)[:, None, None]
    cfg = dict(EJECTING[name], integrator=name, deactivate_on_ejection="1", log_on_ejection="1", rmax="21")
    out = {}
    for mode in ("1", "2", "0"):
        out[mode] = run_logged(S, e, dict(cfg, lane_mode=mode), 60.0)
    (h1, raw1, _), (h2, raw2, _), (h0, raw0, _) = out["1"], out["2"], out["0"]
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all() and (getattr(h1, k) == getattr(h0, k)).all(), k
    r1, r2 = split(raw1), split(raw2)
    assert r1 == r2 == split(raw0)
    recs = O.parse_log(raw2)
    gone = h2.state == O.DISABLED
    # (rkck calls its monitor once more at the start of every launch, for every system whatever its state --
    # rkck_integrator.hpp:140 -- so a system already disabled is logged again by the next launch)
    assert 0 < gone.sum() < nsys and (len(recs) >= gone.sum() if name.startswith("rkck") else len(recs) == gone.sum())
    assert sorted({r["sys"] for r in recs}) == sorted(int(i) for i in h2.id[gone])
    assert all(r["state"] == O.DISABLED for r in recs)
    for r in recs:                                      # the record is the system as it was stored
        i = r["sys"]
        assert r["time"] == h2.time[i]
        if name == "mvs":       # logged in standard coordinates, converted back, and once more at the end of the launch
            np.testing.assert_allclose(r["bodies"][:, :6], np.hstack([h2.pos[i], h2.vel[i]]), rtol=1e-12, atol=1e-15)
        else:
            assert (r["bodies"][:, :3] == h2.pos[i]).all() and (r["bodies"][:, 3:6] == h2.vel[i]).all()


def test_log_overflow_in_the_lane_per_body_kernel(gpu, oracle):
    """A log buffer too small for the records: the ones that do not fit are dropped and counted, nothing is written
    past the end, the integration itself is unaffected."""
    S, O = gpu, oracle
    e = perturbed(O, 200, 4, seed=5)
    cfg = {"integrator": "hermite_cpu_log", "time_step": "0.001", "log_interval": "0.002", "log_writer": "null"}
    big = run_logged(S, e, dict(cfg, lane_mode="2"), 0.05)
    small = run_logged(S, e, dict(cfg, lane_mode="2"), 0.05, cap=64 << 10)
    for k in FIELDS:
        assert (getattr(big[0], k) == getattr(small[0], k)).all(), k
    nbig, nsmall = len(split(big[1])), len(split(small[1]))
    assert nbig == 200 * 25 and big[2] == 0
    assert 0 < nsmall == (64 << 10) // 288 and small[2] == nbig - nsmall      # what fits, and the rest counted
    assert set(split(small[1])) <= set(split(big[1]))


COMPOSITE = {
    # plugin: (cfg, oracle integrator, oracle params)
    "hermite_adap_close_encounter": ({"time_step_factor": "0.017", "min_time_step": "1e-7"}, "hermite_adap",
                                     dict(time_step_factor=0.017, min_time_step=1e-7)),
    "hermite_cpu_ejection_or_close_encounter": ({"time_step": "0.004"}, "hermite_cpu", dict(time_step=0.004)),
    "rkck_adaptive_close_encounter": ({"min_time_step": "1e-6", "max_time_step": "0.02", "error_tolerance": "1e-20"},
                                      "rkck_adaptive", dict(min_time_step=1e-6, max_time_step=0.02, error_tolerance=1e-20)),
    "mvs_close_encounter": ({"time_step": "0.004"}, "mvs", dict(time_step=0.004)),
}


@pytest.mark.parametrize("name", sorted(COMPOSITE))
@pytest.mark.parametrize("logged", [False, True])
def test_ejection_or_close_encounter_composite(gpu, oracle, name, logged):
    """stop_on_ejection_or_close_encounter (composites.hpp:145-222) in the lane-per-body kernels: head-on
    planets that meet at different times plus a few fast escapers; with and without log_on_* records.  States, stop
    times, coordinates and records equal the lane-per-system kernel's; states equal the oracle's."""
    from conftest import head_on
    S, O = gpu, oracle
    e = head_on(O, 128)
    e.vel[5::9, 2] *= 1.6                               # some second planets leave instead: the ejection half
    kw, oname, okw = COMPOSITE[name]
    cfg = dict(kw, integrator=name, deactivate_on_close_encounter="1", close_approach="1.5",
               deactivate_on_ejection="1", rmax="30")
    if logged:
        cfg.update(log_on_close_encounter="1", log_on_ejection="1")
    out = {}
    for mode in ("1", "2", "0"):
        out[mode] = run_logged(S, e, dict(cfg, lane_mode=mode), 8.0)
    (h1, raw1, _), (h2, raw2, _), (h0, raw0, _) = out["1"], out["2"], out["0"]
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all() and (getattr(h1, k) == getattr(h0, k)).all(), k
    assert split(raw1) == split(raw2) == split(raw0)
    gone = h2.state == O.DISABLED
    assert 20 < gone.sum() < 128 and len(np.unique(h2.time[gone])) > 10
    recs = O.parse_log(raw2)
    if logged:
        assert sorted({r["sys"] for r in recs}) == sorted(int(i) for i in h2.id[gone])
    else:
        assert recs == []
    o = e.clone()
    O.integrate(o, O.params(oname, 8.0, **okw),
                O.monitor(kind=O.MON_EJECTION_OR_CLOSE, rmax=30.0, deactivate_on_ejection=1, close_approach=1.5,
                          deactivate_on_close_encounter=1, log_on_close_encounter=int(logged), log_on_ejection=int(logged)))
    assert (h2.state == o.state).all()
    if not name.startswith("rkck"):                     # rkck's step sizes come through pow(): stop times to a step
        np.testing.assert_allclose(h2.time, o.time, rtol=1e-9)


@pytest.mark.parametrize("which", ["MVS", "RKCK_ADAPTIVE", "IRK2"])
def test_log_time_interval_with_the_other_integrators(gpu, oracle, which, monkeypatch):
    """The C-ABI takes any (integrator, monitor) pair; the reference registers log_time_interval with the Hermite
    integrators only.  Through the ABI (a plugin entry added for the test) the lane-per-body MVS, Cash-Karp and Gauss
    IRK kernels log like the lane-per-system ones: same records, same ensembles -- MVS converting the system that
    logs to standard coordinates and back each time."""
    S, O = gpu, oracle
    import swarm_amd.integrator as I
    monkeypatch.setitem(I.PLUGINS, "integrator_log_test", (getattr(S.capi, which), S.capi.MON_LOG_INTERVAL))
    cfg = {"MVS": {"time_step": "0.004"},
           "RKCK_ADAPTIVE": {"min_time_step": "1e-6", "max_time_step": "0.01", "error_tolerance": "1e-20"},
           "IRK2": {"time_step": "0.02"}}[which]
    e = perturbed(O, 150, 5, seed=41)
    e.time[:] = np.linspace(0.0, 0.05, 150)
    out = {}
    for mode in ("1", "2"):
        out[mode] = run_logged(S, e, dict(cfg, integrator="log_test", log_interval="0.0503", lane_mode=mode), 0.6)
    (h1, raw1, _), (h2, raw2, _) = out["1"], out["2"]
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all(), k
    r1, r2 = split(raw1), split(raw2)
    assert r1 == r2 and len(r2) >= 150 * 10
