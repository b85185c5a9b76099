"""The reference's Python unit tests (py/tests/trivial.py + abstract.py) against the
swarmng-compatible module, with the reference's own configuration (integrator=hermite_cpu,
nogpu=1): the hermite_cpu name is served by the HIP Hermite kernel, nogpu is accepted."""
import math
import random

import numpy as np
import pytest


def fill(list_like, initial_value):
    for i in range(0, len(list_like)):
        list_like[i] = initial_value


def make_test_case(swarmng, nsys=16, nbod=3, spacing_factor=1.4, planet_mass=0.001, ejection_factor=1, seed=None):
    """py/tests/trivial.py:18-42, verbatim semantics"""
    random.seed(seed)
    d = swarmng.DefaultEnsemble.create(nbod, nsys)
    for i in range(0, d.nsys):
        s = d[i]
        s.id = i
        s.set_active()
        s.time = 0
        s[0].pos = [0, 0, 0]
        s[0].vel = [0, 0, 0]
        s[0].mass = 1
        fill(s.attributes, 0)
        for j in range(0, d.nbod):
            fill(s[j].attributes, 0)
        for j in range(1, d.nbod):
            r = spacing_factor ** (j - 1)
            v = math.sqrt(1 / r) * ejection_factor
            phi = random.uniform(0, 2 * math.pi)
            s[j].pos = [r * math.cos(phi), r * math.sin(phi), 0]
            s[j].vel = [-v * math.sin(phi), v * math.cos(phi), 0]
            s[j].mass = planet_mass
    return d


def test_api_surface_without_gpu(tmp_path):
    import swarm_amd.swarmng as swarmng
    cfg = swarmng.config({"nsys": 4}, nbod=3, integrator="hermite", time_step=1e-3)
    assert cfg["nsys"] == "4" and cfg["time_step"] == "0.001"
    e = swarmng.generate_ensemble(cfg)
    assert len(e) == 4 and e.nbod == 3 and len(e[0]) == 3
    s = e[1]
    assert s.is_active() and s.id == 1 and s[0].mass == 1.0 and s[1].distance_to_origin() == pytest.approx(10.0)
    s[2][0].pos = 3.5
    assert e.pos[1, 2, 0] == 3.5 and s[2].pos[0] == 3.5
    s.set_disabled()
    assert s.state == -1 and not s.is_enabled()
    with pytest.raises(IndexError):
        e[4]
    c = e.clone()
    c[0].time = 7.0
    assert e[0].time == 0.0
    e.save_to_text(str(tmp_path / "a.txt"))
    e.save_to_bin(str(tmp_path / "a.bin"))
    a, b = swarmng.DefaultEnsemble.load_from_text(str(tmp_path / "a.txt")), swarmng.DefaultEnsemble.load_from_bin(str(tmp_path / "a.bin"))
    assert (b.pos == e.pos).all() and (b.state == e.state).all() and (b.body_attr == e.body_attr).all()
    assert np.abs(a.pos - e.pos).max() < 1e-14 and (a.state == e.state).all()
    assert swarmng.find_max_energy_conservation_error(b, e) == 0.0
    assert e[0].total_energy == pytest.approx(e.calc_total_energy()[0])


def test_calc_keplerian_for_cartesian():
    """module.cpp:186; elements of known orbits (tutorials/kepler.cpp:102-200 conventions)"""
    import swarm_amd.swarmng as swarmng
    a, e, i, O, w, M = swarmng.calc_keplerian_for_cartesian(10.0, 0, 0, 0, math.sqrt(1.0 / 10.0), 0, 1.0)
    assert a == pytest.approx(10.0) and e == 0.0 and i == 0.0 and O == 0.0
    # ellipse a=2, e=0.5 at pericentre (r=1, v^2 = GM(2/r - 1/a) = 1.5), inclined by 30 degrees about the x axis
    v = math.sqrt(1.5)
    a, e, i, O, w, M = swarmng.calc_keplerian_for_cartesian(1.0, 0, 0, 0, v * math.cos(math.pi / 6), v * math.sin(math.pi / 6), 1.0)
    assert a == pytest.approx(2.0) and e == pytest.approx(0.5) and i == pytest.approx(math.pi / 6)
    assert M == pytest.approx(0.0, abs=1e-7) and O == pytest.approx(0.0, abs=1e-12) and w == pytest.approx(0.0, abs=1e-7)
    # hyperbola: v^2 = 2/r + 1/a with a = 1 -> e = 1 + r_p/a = 2 at pericentre r_p = 1
    a, e, i, O, w, M = swarmng.calc_keplerian_for_cartesian(1.0, 0, 0, 0, math.sqrt(3.0), 0, 1.0)
    assert a == pytest.approx(1.0) and e == pytest.approx(2.0) and M == pytest.approx(0.0, abs=1e-7)


def test_keplerian_cartesian_round_trip():
    import swarm_amd.swarmng as swarmng
    rng = random.Random(3)
    for _ in range(50):
        el = [rng.uniform(0.5, 30), rng.uniform(0.01, 0.9), rng.uniform(0.05, 3.0), rng.uniform(0.1, 6.2),
              rng.uniform(0.1, 6.2), rng.uniform(0.1, 6.2)]
        GM = rng.uniform(0.5, 2.0)
        xv = swarmng.calc_cartesian_for_keplerian(*el, GM=GM)
        back = swarmng.calc_keplerian_for_cartesian(*xv, GM)
        for k, (p, q) in enumerate(zip(el, back)):
            d = abs(p - q) if k < 3 else abs((p - q + math.pi) % (2 * math.pi) - math.pi)
            assert d < 1e-7, (k, el, back)


def test_snapshot_files_interoperate_with_cli_and_reference_fixtures(tmp_path, oracle):
    """python text/binary I/O reads the reference's fixture and the C++ CLI's binary output"""
    import os
    import subprocess
    import swarm_amd.swarmng as swarmng
    from conftest import GOLDEN, ROOT
    src = os.path.join(GOLDEN, "integrators", "test.4.out.txt")
    a = swarmng.DefaultEnsemble.load_from_text(src)
    o = oracle.load_text(src)
    for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr"):
        assert (getattr(a, k) == getattr(o, k)).all(), k
    binf = str(tmp_path / "c.bin")
    subprocess.check_call([os.path.join(ROOT, "swarm_amd", "bin", "swarm"), "convert", "-I", src, "-o", binf])
    b = swarmng.DefaultEnsemble.load_from_bin(binf)
    assert (b.pos == a.pos).all() and (b.vel == a.vel).all() and (b.id == a.id).all()
    a.save_to_bin(str(tmp_path / "p.bin"))
    assert open(str(tmp_path / "p.bin"), "rb").read() == open(binf, "rb").read()


class _IntegrationTest:
    """py/tests/abstract.py IntegrationTest: create ensemble, keep a reference copy, integrate, examine"""
    destination_time = 10.0

    def run(self, swarmng):
        self.ens = self.createEnsemble(swarmng)
        self.ref = self.ens.clone()
        swarmng.init(self.cfg(swarmng))
        integ = swarmng.Integrator.create(self.cfg(swarmng))
        integ.ensemble = self.ens
        integ.destination_time = self.destination_time
        integ.integrate()
        swarmng.sync()
        self.examine(swarmng)


@pytest.mark.gpu
def test_basic_integration_energy_conservation():
    """trivial.py:89-99: dt=1e-4, T=10, nsys=16, nbod=3 -> max |dE/E| < 1e-13"""
    import swarm_amd.swarmng as swarmng

    class T(_IntegrationTest):
        def cfg(self, sw):
            return sw.config(integrator="hermite_cpu", time_step=1e-4, nogpu=1)   # trivial.py:76

        def createEnsemble(self, sw):
            return make_test_case(sw, nsys=16, nbod=3, spacing_factor=1.4, seed=14321)

        def examine(self, sw):
            assert sw.find_max_energy_conservation_error(self.ens, self.ref) < 1e-13
            assert all(s.state == 1 and s.time == pytest.approx(10.0, rel=1e-12) for s in self.ens)
    T().run(swarmng)


@pytest.mark.gpu
def test_ejection_test():
    """trivial.py:111-129: dt=1e-3, deactivate_on_ejection=1, rmax=10, T=100, nsys=20 nbod=6,
    spacing 1.01: any body with r > rmax => its system is disabled (state == -1)"""
    import swarm_amd.swarmng as swarmng
    RMAX = 10

    class T(_IntegrationTest):
        destination_time = 100

        def cfg(self, sw):
            return sw.config(integrator="hermite_cpu", time_step=1e-3, nogpu=1, deactivate_on_ejection=1, rmax=RMAX)

        def createEnsemble(self, sw):
            return make_test_case(sw, nsys=20, nbod=6, spacing_factor=1.01, seed=99)

        def examine(self, sw):
            n_ejected = 0
            for s in self.ens:
                for b in s:
                    if b.distance_to_origin() > RMAX:
                        n_ejected += 1
                        assert s.state == -1, "a system with an ejected body should be disabled"
            assert sum(1 for s in self.ens if s.state == -1) > 0 or n_ejected == 0
    T().run(swarmng)


@pytest.mark.gpu
def test_tutorial_flows(tmp_path):
    """py/tutorial.py, py/resume_tutorial.py and py/stats_tutorial.py, statement by statement, over the
    reference's own data file test/integrators/test.3.in.txt (tests/golden/integrators)."""
    import os
    import swarm_amd.swarmng as swarmng
    from conftest import GOLDEN
    # ---- tutorial.py
    swarmng.init(swarmng.config(nogpu=1))
    ref = swarmng.DefaultEnsemble.load_from_text(os.path.join(GOLDEN, "integrators", "test.3.in.txt"))
    lines = []
    for s in ref:
        lines.append("System {0}, time: {1}, state: {2} ".format(s.id, s.time, s.state))
        for i, b in enumerate(s):
            lines.append("\tBody {0} pos: {1}\tvel:{2}".format(i, b.pos, b.vel))
    assert len(lines) == 16 * 4
    ref[3][0].pos = [1.0, 2.0, 5.0]
    assert list(ref[3][0].pos) == [1.0, 2.0, 5.0]
    ref[3][0].pos = [0.0, 0.0, 0.0]
    ens = ref.clone()
    integ = swarmng.Integrator.create(swarmng.config(integrator="hermite_cpu", time_step=0.001))
    integ.ensemble = ens
    integ.destination_time = 1.0
    integ.integrate()
    first = swarmng.DefaultEnsemble.load_from_text(os.path.join(GOLDEN, "integrators", "test.3.out.txt"))
    assert max(abs(a - b) for s, r in zip(ens, first) for p, q in zip(s, r) for a, b in zip(p.pos, q.pos)) < 1e-10
    integ.destination_time = 2.0
    integ.integrate()                                  # systems carry their own time: this resumes at t=1
    assert all(abs(s.time - 2.0) < 1e-9 for s in ens)
    assert swarmng.find_max_energy_conservation_error(ens, ref) < 1e-9
    # ---- resume_tutorial.py: snapshot, reload, continue from the median system time
    snap = str(tmp_path / "in.bin.snapshot")
    ens.save_to_bin(snap)
    again = swarmng.DefaultEnsemble.load_from_bin(snap)
    times = sorted(map(lambda s: s.time, again))
    starting_time = times[len(times) // 2]
    integ = swarmng.Integrator.create(swarmng.config(integrator="hermite_cpu", time_step=0.001))
    integ.ensemble = again
    integ.destination_time = starting_time + 0.5
    integ.integrate()
    assert all(abs(s.time - 2.5) < 1e-9 for s in again)
    # ---- stats_tutorial.py: astrocentric elements of every planet through keplerian_for_cartesian
    value = np.empty((again.nbod - 1, 6, again.nsys))
    for sysid, s in enumerate(again):
        star = s[0]
        center = (star.pos, star.vel, star.mass)
        for bodid in range(1, again.nbod):
            bod = s[bodid]
            value[bodid - 1, :, sysid] = swarmng.keplerian_for_cartesian((bod.pos, bod.vel, bod.mass), center)
    r0 = [np.mean([s[b].distance_to_origin() for s in ref]) for b in (1, 2)]      # circular orbits at t=0
    assert abs(np.average(value[0, 0, :]) / r0[0] - 1) < 0.01 and abs(np.average(value[1, 0, :]) / r0[1] - 1) < 0.01
    assert np.average(value[:, 1, :]) < 0.02 and (value[:, 2, :] == 0).all()


@pytest.mark.gpu
def test_benchmark_script_flow():
    """py/benchmark.py: one `hermite` integrator object, re-used for generated ensembles of 8000 systems with
    3..6 bodies (the device ensemble is re-created when the shape changes), T=10, dt=1e-3"""
    import swarm_amd.swarmng as swarmng
    swarmng.init(swarmng.config(CUDA_DEVICE=0, verbose=1))
    integ = swarmng.Integrator.create(swarmng.config(integrator="hermite", time_step=0.001))
    for nb in range(3, 7):
        ens = swarmng.generate_ensemble(swarmng.config(nsys=8000, nbod=nb))
        ref = ens.clone()
        integ.ensemble = ens
        integ.destination_time = 10.0
        integ.integrate()
        assert ens.nbod == nb and all(abs(s.time - 10.0) < 1e-9 and s.state == 1 for s in ens)
        assert swarmng.find_max_energy_conservation_error(ens, ref) < 1e-7
