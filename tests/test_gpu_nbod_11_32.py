"""nbod 11 .. 32: one lane per body only (SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM = 10 < nbod <= SWARM_HIP_MAX_NBOD = 32).
The reference instantiates nbod = 3 .. MAX_NBODIES and accepts MAX_NBODIES up to 32 (cmake/variables.cmake:55-72), its
ctest loops over that range (cmake/test_integrators.cmake:33-39) and it ships golden pairs up to test.13.{in,out}.txt;
beyond 13 the oracle is the only check.  From nbod 17 on the kernels' loops over a body's partners are rolled and the
masses sit in LDS (BodyLoops, device_common.hpp): 17, 21, 22 and 32 are the ends of the three- and two-systems-per-
wavefront ranges.  Bars: the goldens at the
ctest's 1e-8, the oracle at north_star's 1e-10 with bit-exact times, states and step counts of fixed-step integrators."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, perturbed, rel_diff, to_host
from test_gpu_integrators import CASES, run_gpu

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("nbod", [11, 12, 13])
def test_hermite_vs_reference_golden_files_nbod_11_13(gpu, oracle, nbod):
    """Hermite.cfg (time_step=0.01) vs test.N.out.txt at 1e-8, and vs the oracle's hermite / hermite_cpu at 1e-10"""
    S, O = gpu, oracle
    e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % nbod))
    ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.out.txt" % nbod))
    h = to_host(S, e)
    run_gpu(S, h, "hermite", {"time_step": "0.01"}, 1.0, counters=False)
    pd = np.sqrt(((h.pos - ref.pos) ** 2).sum(-1)).max()
    vd = np.sqrt(((h.vel - ref.vel) ** 2).sum(-1)).max()
    assert pd < 1e-8 and vd < 1e-8, (pd, vd)
    assert (h.state == ref.state).all() and (h.id == ref.id).all()
    for form in ("hermite", "hermite_cpu"):
        o = e.clone()
        O.integrate(o, O.params(form, 1.0, time_step=0.01))
        assert rel_diff(h.pos[:, 1:], o.pos[:, 1:]) < TOL and rel_diff(h.vel[:, 1:], o.vel[:, 1:]) < TOL
        assert np.abs(h.pos[:, 0] - o.pos[:, 0]).max() < TOL and np.abs(h.vel[:, 0] - o.vel[:, 0]).max() < TOL
        assert (h.time == o.time).all() and (h.state == o.state).all()


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("nbod", [11, 13, 16, 17, 21, 22, 32])
def test_integrator_vs_oracle_nbod_11_32(gpu, oracle, name, nbod):
    """every integrator on a ragged ensemble (37 systems: 5, 4, 3 or 2 per wavefront, a last wavefront of 2 or 1)"""
    S, O = gpu, oracle
    cfg, okw, T = CASES[name]
    T = 0.5
    e = perturbed(O, 37, nbod, seed=nbod + 20)
    e.time[::5] = 0.125
    e.state[3::11] = O.DISABLED
    h = to_host(S, e)
    integ = run_gpu(S, h, name, cfg, T)
    acc, att = O.integrate(e, O.params(name, T, **okw), threads=4, counters=True)
    assert (h.state == e.state).all() and (h.id == e.id).all()
    assert rel_diff(h.pos[:, 1:], e.pos[:, 1:]) < TOL, rel_diff(h.pos[:, 1:], e.pos[:, 1:])
    assert rel_diff(h.vel[:, 1:], e.vel[:, 1:]) < TOL
    assert np.abs(h.pos[:, 0] - e.pos[:, 0]).max() < TOL and np.abs(h.vel[:, 0] - e.vel[:, 0]).max() < TOL
    gacc, gatt = integ.step_counts()
    if name in ("euler", "midpoint", "mvs", "rkck_fixed", "hermite_prop", "irk2", "irk2_cpu"):
        assert (gacc == acc.astype(np.uint64)).all() and (gatt == att.astype(np.uint64)).all()
        assert (h.time == e.time).all()
    else:
        assert np.abs(gacc.astype(np.int64) - acc).max() <= 1
        np.testing.assert_allclose(h.time, e.time, rtol=1e-12)


@pytest.mark.parametrize("nbod", [12, 16, 19, 32])
def test_monitors_nbod_above_10(gpu, oracle, nbod):
    """stop_on_ejection (deactivate + log) and log_time_interval beyond nbod 10: stop flags, stop times and the
    records' bytes against the oracle"""
    S, O = gpu, oracle
    # every third system: the outermost planet just inside rmax, moving straight out -- it crosses rmax at t ~ 1; the
    # others stay on their circles
    e = perturbed(O, 23, nbod, seed=3, amp=1e-4)
    r0 = 10 * 1.4 ** (nbod - 2)
    r = e.pos[::3, nbod - 1].copy()
    e.pos[::3, nbod - 1] = r * 1.045
    e.vel[::3, nbod - 1] = r / np.sqrt((r ** 2).sum(-1))[:, None] * (0.005 * r0)
    h = to_host(S, e)
    rmax = 1.05 * r0
    cfg = {"integrator": "hermite", "time_step": "0.01", "deactivate_on_ejection": "1", "log_on_ejection": "1",
           "rmax": repr(rmax)}
    integ = S.Integrator.create(S.config(cfg))
    lm = S.LogManager(1 << 20, "host")
    integ.set_log_manager(lm)
    integ.set_ensemble(h)
    integ.set_destination_time(3.0)
    integ.integrate()
    olog = O.Log(1 << 20)
    O.integrate(e, O.params("hermite", 3.0, time_step=0.01), O.monitor(rmax=rmax, deactivate_on_ejection=1, log_on_ejection=1),
                olog)
    assert (h.state == e.state).all() and (h.state == O.DISABLED).sum() == len(range(0, 23, 3))
    assert (h.time == e.time).all() and 0.5 < h.time[0] < 2.0 and abs(h.time[1] - 3.0) < 1e-12
    mine = sorted(O.parse_log(bytes(lm.host)), key=lambda r: r["sys"])
    theirs = sorted(olog.records(), key=lambda r: r["sys"])
    assert len(mine) == len(theirs) == (h.state == O.DISABLED).sum()
    for a, b in zip(mine, theirs):
        assert a["sys"] == b["sys"] and a["time"] == b["time"] and a["state"] == b["state"] and a["nbod"] == nbod
        assert rel_diff(a["bodies"][1:, :3], b["bodies"][1:, :3]) < TOL
    # log_time_interval: record counts per system
    e = perturbed(O, 9, nbod, seed=4)
    h = to_host(S, e)
    integ = S.Integrator.create(S.config({"integrator": "hermite_adap_log", "time_step_factor": "0.02",
                                          "min_time_step": "1e-7", "log_interval": "0.1"}))
    lm = S.LogManager(4 << 20, "host")
    integ.set_log_manager(lm)
    integ.set_ensemble(h)
    integ.set_destination_time(0.5)
    integ.integrate()
    olog = O.Log(4 << 20)
    O.integrate(e, O.params("hermite_adap", 0.5, time_step_factor=0.02, min_time_step=1e-7),
                O.monitor(kind=O.MON_LOG_INTERVAL, log_interval=0.1), olog)
    mine, theirs = O.parse_log(bytes(lm.host)), olog.records()
    assert len(mine) == len(theirs) and lm.dropped == 0
    assert np.array_equal(np.bincount([r["sys"] for r in mine], minlength=9), np.bincount([r["sys"] for r in theirs], minlength=9))
    assert len(bytes(lm.host)) == len(mine) * ((32 + 64 * nbod + 15) // 16 * 16)


@pytest.mark.parametrize("nbod", [17, 22, 32])
@pytest.mark.parametrize("nsys", [1, 2, 3])
def test_fewer_systems_than_a_wavefront_holds(gpu, oracle, nbod, nsys):
    """one, two or three systems where a wavefront holds three or two: idle groups of lanes, a single workgroup"""
    S, O = gpu, oracle
    for name, cfg, okw in (("hermite", {"time_step": "0.01"}, dict(time_step=0.01)),
                           ("mvs", {"time_step": "0.01"}, dict(time_step=0.01)),
                           ("rkck_adaptive", {"min_time_step": "1e-4", "max_time_step": "0.01", "error_tolerance": "1e-12"},
                            dict(min_time_step=1e-4, max_time_step=0.01, error_tolerance=1e-12))):
        e = perturbed(O, nsys, nbod, seed=nbod + nsys)
        h = to_host(S, e)
        run_gpu(S, h, name, cfg, 0.3)
        O.integrate(e, O.params(name, 0.3, **okw))
        assert (h.state == e.state).all() and (h.state == O.INACTIVE).all()
        assert rel_diff(h.pos[:, 1:], e.pos[:, 1:]) < TOL and rel_diff(h.vel[:, 1:], e.vel[:, 1:]) < TOL, name
        np.testing.assert_allclose(h.time, e.time, rtol=1e-12)


def head_of(O, e, k):
    """the first k systems of an ensemble as an ensemble of their own"""
    h = O.generate(k, e.nbod)
    for f in ("pos", "vel", "mass", "time", "state", "id"):
        getattr(h, f)[...] = getattr(e, f)[:k]
    return h


@pytest.mark.parametrize("name,cfg,nbod,T", [
    ("hermite", {"time_step": "0.01"}, 12, 0.08), ("hermite", {"time_step": "0.01"}, 16, 0.08),
    ("hermite_adap", {"time_step_factor": "0.02", "min_time_step": "1e-6"}, 13, 0.3),
    ("hermite", {"time_step": "0.01", "deactivate_on_ejection": "1", "rmax": "1e9"}, 15, 0.08),
    ("mvs", {"time_step": "0.01"}, 14, 0.08), ("mvs", {"time_step": "0.01"}, 16, 0.08),
    ("mvs", {"time_step": "0.01", "deactivate_on_ejection": "1", "rmax": "1e9"}, 15, 0.08),
    ("euler", {"time_step": "0.001"}, 13, 0.008), ("midpoint", {"time_step": "0.001"}, 16, 0.008),
    ("hermite_prop", {"time_step": "0.001"}, 14, 0.008),
    ("hermite_prop", {"time_step": "0.001", "deactivate_on_ejection": "1", "rmax": "1e9"}, 16, 0.008),
])
def test_both_loop_forms_give_the_same_bits(gpu, oracle, name, cfg, nbod, T):
    """nbod 12 .. 16, Hermite, MVS, euler, midpoint, hermite_prop: the loops over a body's partners are unrolled for a small ensemble and rolled
    (masses in LDS) for a large one (BodyLoops, device_common.hpp; the launcher goes by the number of systems).  The
    first 90 systems of 16500 must come out bit for bit as the same 90 systems integrated alone."""
    S, O = gpu, oracle
    big = perturbed(O, 16500, nbod, seed=nbod)
    big.time[::7] = 0.01
    big.state[5::13] = O.DISABLED
    small = head_of(O, big, 90)
    hb, hs = to_host(S, big), to_host(S, small)
    ib = run_gpu(S, hb, name, cfg, T)
    isml = run_gpu(S, hs, name, cfg, T)
    for f in ("pos", "vel", "time", "state"):
        assert (getattr(hb, f)[:90] == getattr(hs, f)).all(), f
    assert (ib.step_counts()[0][:90] == isml.step_counts()[0]).all()
    # and the small one against the oracle, as everywhere
    O.integrate(small, O.params(name, T, **{k: float(v) for k, v in cfg.items() if k in ("time_step", "time_step_factor", "min_time_step")}),
                O.monitor(rmax=1e9, deactivate_on_ejection=1) if "rmax" in cfg else None)
    assert rel_diff(hs.pos[:, 1:], small.pos[:, 1:]) < TOL and rel_diff(hs.vel[:, 1:], small.vel[:, 1:]) < TOL
    assert (hs.state == small.state).all()


@pytest.mark.parametrize("nbod,name,cfg,okw", [
    (12, "hermite", {"time_step": "0.01"}, dict(time_step=0.01)),
    (19, "hermite_adap", {"time_step_factor": "0.02", "min_time_step": "1e-7"}, dict(time_step_factor=0.02, min_time_step=1e-7)),
    (14, "mvs", {"time_step": "0.01"}, dict(time_step=0.01)),
    (32, "rkck_adaptive", {"min_time_step": "1e-4", "max_time_step": "0.01", "error_tolerance": "1e-12"},
     dict(min_time_step=1e-4, max_time_step=0.01, error_tolerance=1e-12)),
])
def test_verbose_on_ejection_beyond_nbod_10(gpu, oracle, nbod, name, cfg, okw):
    """verbose_on_ejection=1 where only lane-per-body kernels exist: the lane of the first planet that meets the test
    prints the reference's line (stop_on_ejection.hpp:120-130, 184-185); lines, stop times and records vs the oracle.
    In every third system the two outermost planets leave together: only the inner one of them is reported."""
    from test_gpu_mce import run, texts
    S, O = gpu, oracle
    e = perturbed(O, 23, nbod, seed=3, amp=1e-4)
    r0 = 10 * 1.4 ** (nbod - 2)
    for b in (nbod - 2, nbod - 1):
        r = e.pos[::3, b].copy()
        e.pos[::3, b] = r / np.sqrt((r ** 2).sum(-1))[:, None] * (1.045 * r0)
        e.vel[::3, b] = r / np.sqrt((r ** 2).sum(-1))[:, None] * (0.005 * r0)
    rmax = 1.05 * r0
    cfg = dict(cfg, integrator=name, rmax=repr(rmax), deactivate_on_ejection="1", verbose_on_ejection="1", log_on_ejection="1")
    h, lm, o, olog = run(S, O, e, cfg, 3.0, okw,
                         O.monitor(rmax=rmax, deactivate_on_ejection=1, verbose_on_ejection=1, log_on_ejection=1))
    assert (h.state == o.state).all() and (o.state == O.DISABLED).sum() == len(range(0, 23, 3))
    gt, gs = texts(O, lm.host)
    ot, os_ = texts(O, olog.bytes())
    # (rkck calls its monitor before the loop of EVERY launch for every system, rkck_integrator.hpp:140: a second
    # launch reports the ejected systems again, in the reference as here)
    assert len(gt) == len(ot) and len(gs) == len(os_) and len(gt) == (16 if name == "rkck_adaptive" else 8), (len(gt), len(ot))
    strip = lambda t: t.split(" T=")[0]                    # "_sys=%d, bod=%d" part is exact
    assert sorted(map(strip, gt)) == sorted(map(strip, ot))
    assert all("bod=%d," % (nbod - 2) in t for t in gt)
    if name in ("hermite", "hermite_adap"):
        assert gt == ot                                    # %lg prints 6 digits: identical text


@pytest.mark.parametrize("nbod", [11, 24])
def test_verbose_on_close_encounter_beyond_nbod_10(gpu, oracle, nbod):
    """verbose_on_close_encounter=1: the pairs (b, d < b) are tested in order until one is met
    (stop_on_close_encounter.hpp:108-110, 137-141); planet 5 starts next to planet 4 in every other system, and
    planet 8 next to planet 3 as well in every fourth -- the pair (5, 4) comes first"""
    from test_gpu_mce import run, texts
    S, O = gpu, oracle
    e = perturbed(O, 20, nbod, seed=8, amp=1e-4)
    e.pos[::2, 5] = e.pos[::2, 4] * 1.002
    e.pos[::4, 8] = e.pos[::4, 3] * 1.002
    cfg = {"integrator": "hermite_adap_close_encounter", "time_step_factor": "0.017", "min_time_step": "1e-7",
           "close_approach": "2.0", "deactivate_on_close_encounter": "1", "verbose_on_close_encounter": "1"}
    h, lm, o, olog = run(S, O, e, cfg, 0.5, dict(time_step_factor=0.017, min_time_step=1e-7),
                         O.monitor(kind=O.MON_EJECTION_OR_CLOSE, close_approach=2.0, deactivate_on_close_encounter=1,
                                   verbose_on_close_encounter=1))
    assert (h.state == o.state).all() and (o.state == O.DISABLED).sum() == 10
    gt, _ = texts(O, lm.host)
    ot, _ = texts(O, olog.bytes())
    assert gt == ot and len(gt) == 10
    assert all("j=4 i=5" in t for t in gt)


def test_what_stops_at_nbod_10_says_so(gpu, oracle):
    """one lane per system, mce_stat and log_transit stop at nbod 10: a clear error, not a fallback"""
    S, O = gpu, oracle
    h = to_host(S, perturbed(O, 8, 11, seed=1))
    for cfg in ({"integrator": "hermite", "time_step": "0.01", "lane_mode": "1"},
                {"integrator": "hermite_adap_transit", "time_step_factor": "0.02", "min_time_step": "1e-6", "nbod": "11", "time_step": "0.01"}):
        integ = S.Integrator.create(S.config(cfg))
        integ.set_ensemble(h.clone())
        integ.set_destination_time(0.1)
        if "lane_mode" in cfg:
            integ.integrate()          # lane_mode=1 cannot be honoured beyond nbod 10: one lane per body is what runs
            continue
        with pytest.raises(S.capi.SwarmHipError, match="one lane per body"):
            integ.integrate()
    h33 = S.HostEnsemble(4, 33)
    with pytest.raises(S.capi.SwarmHipError, match="nbod"):
        S.DeviceEnsemble(4, 33).upload(h33)
