"""Pins the CPU oracle against the reference's own golden vectors (no GPU).

tests/golden/integrators/test.N.{in,out}.txt are the reference's files
(test/integrators/, produced by `hermite_cpu time_step=0.001 destination_time=1`,
test/integrators/generate.sh + Hermite_CPU.cfg).  The text format carries 16 significant
digits, so agreement is bounded by text rounding of O(|x| * 1e-16)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.mark.parametrize("nbod", range(3, 14))          # every test.N.{in,out}.txt the reference ships
def test_hermite_cpu_reproduces_reference_golden(oracle, nbod):
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % nbod))
    ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.out.txt" % nbod))
    O.integrate(e, O.params("hermite_cpu", 1.0, time_step=0.001))
    pd, vd, td = O.compare(e, ref)
    scale = np.abs(ref.pos).max()
    assert pd <= 4e-16 * scale * 4, pd      # text rounding only
    assert vd <= 1e-15, vd
    assert td == 0.0
    assert (e.state == ref.state).all() and (e.state == O.INACTIVE).all()
    assert (e.id == ref.id).all()


@pytest.mark.parametrize("nbod", [3, 4, 6, 9])
def test_generator_matches_reference_inputs(oracle, nbod):
    """generate_ensemble (utils.cpp:63-103) vs the shipped `swarm generate` outputs"""
    O = oracle
    g = O.generate(16, nbod)
    e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % nbod))
    assert np.abs(g.pos - e.pos).max() <= 4e-16 * np.abs(e.pos).max() * 2
    assert np.abs(g.vel - e.vel).max() <= 1e-16
    assert (g.mass == e.mass).all() and (g.body_attr == e.body_attr).all()
    assert (g.state == e.state).all() and (g.time == e.time).all() and (g.id == e.id).all()


# ctest bar of the reference: every integrator vs the hermite_cpu goldens at 1e-8
# (cmake/test_integrators.cmake:33-39 with test/integrators/<Id>.cfg).
CFGS = {
    "hermite": dict(time_step=0.01),
    "hermite_adap": dict(time_step_factor=0.02, min_time_step=1e-7),
    "rkck_adaptive": dict(min_time_step=1e-4, max_time_step=1e-3, error_tolerance=1e-27),
    "rkck_fixed": dict(min_time_step=1e-4, max_time_step=1e-4, error_tolerance=1e-27),
    "euler": dict(time_step=1e-5),
    "midpoint": dict(time_step=1e-5),
    "hermite_prop": dict(time_step=0.01),     # samples/Hermite_Prop.cfg family
    "verlet": dict(max_timestep=1e-6, timestep_scale=1.0),     # test/integrators/Verlet.cfg: 10^6 steps per system
}


@pytest.mark.parametrize("name", sorted(CFGS))
@pytest.mark.parametrize("nbod", [3, 4, 5, 6])
def test_gpu_form_restatements_meet_reference_ctest_bar(oracle, name, nbod):
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % nbod))
    ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.out.txt" % nbod))
    O.integrate(e, O.params(name, 1.0, **CFGS[name]), threads=8 if name == "verlet" else 4)
    pd, vd, td = O.compare(e, ref)
    assert pd < 1e-8 and vd < 1e-8 and td < 1e-8, (pd, vd, td)
    assert (e.state == O.INACTIVE).all()


# MVS.cfg (time_step=0.0003, T=1): compare_ensembles distances (pos, vel) of the restatement of propagators/{mvs,
# keplerian}.hpp from test.N.out.txt, measured; the HIP kernel reproduces them to 4e-12 (tests/test_gpu_integrators.py)
MVS_OFFSET = {3: (3.09e-9, 6.02e-9), 4: (4.88e-9, 9.81e-9), 5: (7.29e-9, 1.46e-8), 6: (1.60e-8, 3.21e-8),
              7: (2.95e-8, 5.90e-8), 8: (4.59e-8, 9.19e-8)}


@pytest.mark.parametrize("nbod", [3, 4, 5, 6, 7, 8])
def test_mvs_restatement_vs_golden(oracle, nbod):
    """The reference forms dfdt = (f*dgdt - 1)/g from f and dgdt rounded to doubles next to 1 (keplerian.hpp:214-221):
    a DETERMINISTIC offset of ~1e-16/time_step per step, the same sign every step, growing as 1/time_step^2 -- not
    noise between builds (this restatement compiled with and without FMA contraction agrees with itself to 4e-12).
    It is above the reference's own 1e-8 ctest bar from nbod 5 on.  The second half pins everything ELSE in the MVS
    restatement (coordinates, drift, kick, the Laguerre solver, the Stumpff functions): with the same f, g expressions
    evaluated from f-1 and dg/dt-1 (kepler_compensated, no reference counterpart) MVS agrees with the hermite_cpu
    goldens to 1e-12."""
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % nbod))
    ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.out.txt" % nbod))
    c = e.clone()
    O.integrate(e, O.params("mvs", 1.0, time_step=0.0003))
    pd, vd, td = O.compare(e, ref)
    fp, fv = MVS_OFFSET[nbod]
    assert 0.97 * fp < pd < 1.03 * fp and 0.97 * fv < vd < 1.03 * fv and td == 0.0, (pd, vd)
    O.integrate(c, O.params("mvs", 1.0, time_step=0.0003, kepler_compensated=1))
    pd, vd, td = O.compare(c, ref)
    assert pd < 1e-12 and vd < 1e-13 and td == 0.0, (pd, vd)


def test_mvs_offset_grows_as_inverse_square_of_the_step(oracle):
    """halving time_step quadruples the offset (rounding, not truncation, which would shrink 4x)"""
    O = oracle
    ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.6.out.txt"))
    vd = []
    for dt in (6e-4, 3e-4, 1.5e-4):
        e = O.load_text(os.path.join(GOLDEN, "integrators", "test.6.in.txt"))
        O.integrate(e, O.params("mvs", 1.0, time_step=dt), threads=4)
        vd.append(O.compare(e, ref)[1])
    assert 3.0 < vd[1] / vd[0] < 5.0 and 3.0 < vd[2] / vd[1] < 5.0, vd


def test_legacy_sample_fixture(oracle):
    """test/sample/{in,out}.txt + CPU.cfg (hermite_cpu dt=1e-4, T=1).  The scenario is commented
    out in the reference's ctest (cmake/test.cmake:30) and the output predates the current end-of-
    integration rule (its times are exactly 1.0), so it pins positions to 1e-12 only."""
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "sample", "in.txt"))
    ref = O.load_text(os.path.join(GOLDEN, "sample", "out.txt"))
    O.integrate(e, O.params("hermite_cpu", 1.0, time_step=1e-4))
    pd, vd, td = O.compare(e, ref)
    assert pd < 1e-12 and vd < 1e-12 and td < 1e-12


def test_pair_map_covers_all_pairs(oracle):
    """src/utils/unit_tests.cpp:39-75: first/second<nbod>(ij) enumerates every unordered pair once"""
    O = oracle
    assert [O.pair(3, ij) for ij in range(3)] == [(2, 0), (1, 0), (2, 1)]
    for nbod in range(3, O.MAX_NBOD + 1):
        P = nbod * (nbod - 1) // 2
        pairs = {O.pair(nbod, ij) for ij in range(P)}
        assert len(pairs) == P
        assert all(0 <= b < a < nbod for a, b in pairs)


@pytest.mark.parametrize("nbod", [14, 17, 24, 32])
def test_restatements_beyond_the_golden_range(oracle, nbod):
    """nbod 14 .. 32 (cmake/variables.cmake:55-72 accepts MAX_NBODIES up to 32; the goldens stop at 13): hermite_cpu
    on generated systems conserves energy as it does below 13, and the GPU-form restatements agree with it to the bar
    they meet against the goldens"""
    O = oracle
    assert O.MAX_NBOD == 32
    g = O.generate(4, nbod)
    e0 = O.total_energy(g)
    ref = g.clone()
    O.integrate(ref, O.params("hermite_cpu", 1.0, time_step=0.01))
    assert np.abs((O.total_energy(ref) - e0) / e0).max() < 1e-9
    for name, kw in (("hermite", {"time_step": 0.01}), ("hermite_prop", {"time_step": 0.01})):
        e = g.clone()
        O.integrate(e, O.params(name, 1.0, **kw))
        pd, vd, td = O.compare(e, ref)
        assert pd < 1e-10 and vd < 1e-10 and td < 1e-12, (name, pd, vd, td)


def test_snapshot_record_layout(oracle):
    """224 B (nbod=3) / 288 B (nbod=4) records; byte layout of one record"""
    O = oracle
    assert O.lib().orc_snapshot_record_size(3) == 224
    assert O.lib().orc_snapshot_record_size(4) == 288
    e = O.generate(2, 3)
    log = O.Log(1 << 16)
    O.integrate(e, O.params("hermite_cpu", 0.05, time_step=0.01),
                O.monitor(kind=O.MON_LOG_INTERVAL, log_interval=0.02), log)
    recs = log.records()
    assert len(recs) == 2 * 3 and all(r["len"] == 224 and r["msgid"] == 1 for r in recs)
    assert all((r["body_id"] == np.arange(3)).all() for r in recs)
    assert sorted(round(r["time"], 12) for r in recs if r["sys"] == 0) == [0.01, 0.02, 0.04]


def test_bdb_log_count_golden(oracle):
    """test/bdb/bdb.sh: hermite_cpu_log dt=0.01 log_interval=0.1 T=1 on test.4.in.txt ->
    test.4.ref.txt = 528 lines = 16 sys x 3 planets x 11 snapshots; values printed with %9.5lg
    (swarm/query.cpp:193-235)."""
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "bdb", "test.4.in.txt"))
    log = O.Log(1 << 20)
    O.integrate(e, O.params("hermite_cpu", 1.0, time_step=0.01),
                O.monitor(kind=O.MON_LOG_INTERVAL, log_interval=0.1), log)
    recs = log.records()
    assert len(recs) == 16 * 11 and log.c.len == 176 * 288 and log.c.dropped == 0
    ref = [l.split() for l in open(os.path.join(GOLDEN, "bdb", "test.4.ref.txt"))]
    assert len(ref) == 528
    got = {}
    for r in recs:
        for b in range(1, 4):
            got[(float("%lg" % r["time"]), r["sys"], b)] = r["bodies"]
    assert len(got) == 528
    for f in ref:
        t, s, b = float(f[1]), int(f[2]), int(f[3])
        bodies = got[(t, s, b)]
        # `swarm query` default output is Cartesian in JACOBI coordinates: body b relative to the
        # centre of mass of bodies 0..b-1 (swarm/query.cpp:65,95-119,184-185), 5 significant digits
        m = bodies[:b, 6:7]
        center = (bodies[:b, :6] * m).sum(0) / m.sum()
        want = np.array([float(x) for x in f[5:11]])
        assert float(f[4]) == float("%lg" % bodies[b, 6])
        np.testing.assert_allclose(bodies[b, :6] - center, want, rtol=1e-4, atol=1e-5)


def test_hyperbolic_ejection_fixture(oracle):
    """test/monitors/hyperbolic_orbits: hermite_adap, deactivate_on_ejection=1, rmax=20, T=100:
    all 128 systems end DISABLED at t ~ 8 (HermiteEjection.3.out.txt)."""
    O = oracle
    d = os.path.join(GOLDEN, "monitors", "hyperbolic_orbits")
    e = O.load_text(os.path.join(d, "3.in.txt"))
    ref = O.load_text(os.path.join(d, "HermiteEjection.3.out.txt"))
    O.integrate(e, O.params("hermite_adap", 100.0, time_step_factor=0.017, min_time_step=1e-7),
                O.monitor(rmax=20.0, deactivate_on_ejection=1))
    assert (ref.state == O.DISABLED).all()
    assert (e.state == ref.state).all()
    np.testing.assert_allclose(e.time, ref.time, rtol=0, atol=5e-3)


def test_close_orbits_ejection_fixture(oracle):
    """test/monitors/close_orbits: HermiteEjection.cfg on 4.in.txt -> HermiteEjection.4.out.txt.
    These systems are chaotic over T=100 (and one of them spends >10^8 minimum-size steps in a captured pair under
    the oracle's arithmetic), so the run itself is the GPU test's (test_gpu_monitors.py) and only the fixture's
    bulk outcome is recorded here."""
    O = oracle
    d = os.path.join(GOLDEN, "monitors", "close_orbits")
    ref = O.load_text(os.path.join(d, "HermiteEjection.4.out.txt"))
    assert int((ref.state == O.DISABLED).sum()) == 11


def test_close_orbits_close_encounter_fixture(oracle):
    """test/monitors/close_orbits: HermiteCloseEncounter.cfg on 4.in.txt -> HermiteCloseEncounter.4.out.txt, written
    by the reference's GPU `hermite_adap_close_encounter` (T=40, close_approach=0.01, deactivate).  Pins the
    adaptive Hermite restatement together with stop_on_ejection_or_close_encounter: the same 15 systems stop at the
    same times (2e-9), the others end within 2e-8 after T=40 of adaptive steps through near-encounters."""
    O = oracle
    d = os.path.join(GOLDEN, "monitors", "close_orbits")
    e = O.load_text(os.path.join(d, "4.in.txt"))
    ref = O.load_text(os.path.join(d, "HermiteCloseEncounter.4.out.txt"))
    O.integrate(e, O.params("hermite_adap", 40.0, time_step_factor=0.017, min_time_step=1e-7),
                O.monitor(kind=O.MON_EJECTION_OR_CLOSE, close_approach=0.01, deactivate_on_close_encounter=1), threads=4)
    assert (e.state == ref.state).all() and int((ref.state == O.DISABLED).sum()) == 15
    np.testing.assert_allclose(e.time, ref.time, rtol=0, atol=1e-8)
    assert np.abs(e.pos - ref.pos).max() < 1e-7 and np.abs(e.vel - ref.vel).max() < 1e-7
    assert (e.id == ref.id).all()


@pytest.mark.parametrize("name,dt,kw", [("irk2_cpu", 1e-3, {}), ("irk2", 0.01, {}), ("irk2", 0.01, {"method": 4}),
                                        ("irk2", 1e-3, {"method": 2})])
@pytest.mark.parametrize("nbod", [3, 5, 8])
def test_irk2_restatements_meet_the_reference_ctest_bar(oracle, name, dt, kw, nbod):
    """irk2_cpu is enabled by default and has a cfg (test/integrators/irk2_cpu.cfg), so the reference's
    ctest runs it against the hermite_cpu goldens at 1e-8 (cmake/test_integrators.cmake:20-39); irk2.cfg
    does the same for the GPU plugin when it is enabled.  The order-12/8/4 Gauss methods land within
    1e-10 of the goldens."""
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.in.txt" % nbod))
    ref = O.load_text(os.path.join(GOLDEN, "integrators", "test.%d.out.txt" % nbod))
    O.integrate(e, O.params(name, 1.0, time_step=dt, **kw), threads=4)
    pd, vd, td = O.compare(e, ref)
    assert pd < 1e-10 and vd < 1e-12 and td == 0 and (e.state == ref.state).all()
