"""The Python side of the log path (swarm_amd.{logdb,logrecord,range_type,logquery} and the Python `binary`
log writer): mirrors of py/swarmng/{logdb,logrecord,range_type,query}.py served from binary log files.
CPU tests take their record stream from the oracle; the GPU tests take it from the kernels."""
import io
import os
import subprocess
import types

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

SWARM = os.path.join(ROOT, "swarm_amd", "bin", "swarm")


def fixture_log(oracle):
    """test/bdb/bdb.sh's run: hermite_cpu_log dt=0.01 log_interval=0.1 T=1 on test.4.in.txt: 16 x 11 records"""
    O = oracle
    e = O.load_text(os.path.join(GOLDEN, "bdb", "test.4.in.txt"))
    log = O.Log(1 << 20)
    O.integrate(e, O.params("hermite_cpu", 1.0, time_step=0.01), O.monitor(kind=O.MON_LOG_INTERVAL, log_interval=0.1), log)
    return log


def write_binary_log(tmp_path, raw, chunk=1000):
    import swarm_amd as S
    from swarm_amd.integrator import BinaryWriter
    fn = str(tmp_path / "run.log")
    w = BinaryWriter(S.config({"log_output": fn}))
    # records arrive in whatever pieces the flushes deliver; cut on record boundaries like a flush does
    from swarm_amd import gpulog
    recs = [r for _, r in gpulog.records(raw)]
    for i in range(0, len(recs), 7):
        w.process(b"".join(reversed(recs[i:i + 7])))      # out of time order within a flush
    w.close()
    assert not os.path.exists(fn + ".raw")
    return fn


def test_range_type():
    from swarm_amd.range_type import Range, RangeType
    r = Range.interval(5, 8)
    assert r.contains(7) and r.contains(8) and not r.contains(9) and list(r) == [5, 6, 7, 8]
    assert Range.interval((1, 2)).ulPair() == (1, 2)
    s = Range.single(3)
    assert s.isSingle() and s.contains(3) and not s.contains(4) and s.ulPair() == (3, 3) and list(s) == [3]
    u = Range.universal()
    assert u.isUniversal() and u.contains(-1e300) and list(u.with_limits(2, 4)) == [2, 3, 4]
    with pytest.raises(ValueError):
        u.ulPair()
    assert list(r.with_limits(0, 6)) == [5, 6]
    p = RangeType(float)("0.5..2")
    assert p.isInterval() and p.lower() == 0.5 and p.upper() == 2.0
    assert RangeType(int)("7").contains(7)
    assert repr(r) == "[ 5..8]" and repr(u) == "[ .. ]"


def test_binary_writer_file_equals_what_swarm_query_reads(tmp_path, oracle):
    """the Python writer's file: 96-byte swarm_header + records sorted by (time, system); `swarm query` (C++)
    prints the reference's golden table from it"""
    log = fixture_log(oracle)
    fn = write_binary_log(tmp_path, log.bytes())
    raw = open(fn, "rb").read()
    assert raw[:8] == b"SWARM\0" + b"0\0" and raw[8:23] == b"T_sorted_output" and len(raw) == 96 + 176 * 288
    assert int.from_bytes(raw[88:96], "little") == 176 * 288
    p = subprocess.run([SWARM, "query", "-f", fn], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and p.stdout == open(os.path.join(GOLDEN, "bdb", "test.4.ref.txt")).read()


def test_indexed_log_queries(tmp_path, oracle):
    from swarm_amd.logdb import IndexedLogDB, PKey
    from swarm_amd.range_type import Range
    log = fixture_log(oracle)
    want = log.records()
    fn = write_binary_log(tmp_path, log.bytes())
    d = IndexedLogDB(fn)
    assert d.getMetadata("type") == "T_sorted_output" and d.getMetadata("nothing") is None
    allr = list(d.all_records())
    assert len(allr) == 176
    keys = [k.astuple() for k, _ in allr]
    assert keys == sorted(keys) and all(k.event_id == 1 for k, _ in allr)
    # records decode to what the oracle's own parser sees
    by = {(round(r["time"], 9), r["sys"]): r for r in want}
    for k, l in allr:
        r = by[(round(l.time, 9), l.sys)]
        assert l.state == r["state"] and len(l.bodies) == 4 and k.system_id == l.sys
        assert k.time == float(np.float32(l.time))
        got = np.array([list(b.position) + list(b.velocity) + [b.mass] for b in l.bodies])
        assert (got == r["bodies"]).all()
    # primary key <-> the bdb writer's 8-byte key (float time, event << 24 | system)
    k = PKey(0.3, 1, 5)
    assert PKey.fromBinary(k.toBinary()).astuple() == k.astuple() and len(k.toBinary()) == 8
    assert PKey.compareBinary(PKey(0.1, 1, 9).toBinary(), PKey(0.2, 1, 0).toBinary()) == -1
    # time index
    ts = list(d.time_sequence((0.05, 0.55)))
    assert len(ts) == 5 and ts == sorted(ts) and abs(ts[0] - 0.1) < 0.011
    assert len(list(d.time_range_records((0.05, 0.55)))) == 5 * 16
    # system index
    sr = list(d.system_range_records((3, 5)))
    assert len(sr) == 3 * 11 and [k.system_id for k, _ in sr] == [3] * 11 + [4] * 11 + [5] * 11
    assert [k.time for k, _ in sr[:11]] == sorted(k.time for k, _ in sr[:11])
    st = list(d.system_at_time(7, (0.25, 0.75)))
    assert len(st) == 5 and all(k.system_id == 7 and 0.25 <= k.time <= 0.75 for k, _ in st)
    assert len(list(d.system_range_for_time_event(ts[0], 1, (2, 9)))) == 8
    # initial / final conditions
    ic = list(d.initial_conditions(Range.interval(10, 12)))
    fc = list(d.final_conditions(Range.interval(10, 12)))
    assert [s for s, _ in ic] == [10, 11, 12] == [s for s, _ in fc]
    assert all(l.time == min(r["time"] for r in want if r["sys"] == s) for s, l in ic)
    assert all(l.time == max(r["time"] for r in want if r["sys"] == s) for s, l in fc)
    assert len(list(d.initial_conditions(Range.universal()))) == 16
    # combined query, every branch of queryInternal
    U = Range.universal()
    assert len(list(d.query(U, U, U))) == 176
    assert len(list(d.query(U, Range.interval(0, 1), U))) == 22
    assert len(list(d.query(Range.interval(0.05, 0.35), U, U))) == 48
    q = list(d.query(Range.interval(0.05, 0.35), Range.single(4), Range.single(1)))
    assert len(q) == 3 and all(k.system_id == 4 for k, _ in q)
    assert list(d.query(U, U, Range.single(15))) == []
    with pytest.raises(RuntimeError):
        bad = tmp_path / "bad.log"
        bad.write_bytes(b"x" * 200)
        IndexedLogDB(str(bad))


def test_python_query_table_and_keplerian_agree_with_swarm_query(tmp_path, oracle):
    """python -m swarm_amd.logquery against the C++ `swarm query` on the same file: origin-frame Cartesian
    table and Jacobi Keplerian elements (py/swarmng/query.py vs swarm/query.cpp)"""
    from swarm_amd import logquery
    from swarm_amd.range_type import Range
    fn = write_binary_log(tmp_path, fixture_log(oracle).bytes())
    U = Range.universal()

    def args(**kw):
        a = dict(database=fn, max_records=10000, system_range=U, time_range=U, body_range=U, evt_id=U, keplerian=None,
                 initial_conditions=False, final_conditions=False)
        a.update(kw)
        return types.SimpleNamespace(**a)

    def cxx(*opts):
        p = subprocess.run([SWARM, "query", "-f", fn] + list(opts), capture_output=True, text=True, timeout=60)
        assert p.returncode == 0, p.stderr
        return [l.split() for l in p.stdout.splitlines() if l.strip()]

    out = io.StringIO()
    logquery.run_with_args(args(), out)
    rows = [l.split() for l in out.getvalue().splitlines()]
    ref = cxx("--origin", "-b", "0..3")
    assert len(rows) == 176 * 4 == len(ref)
    for a, b in zip(rows, ref):         # msgid time sys body mass x y z vx vy vz state
        assert a[:4] == b[:4]
        np.testing.assert_allclose([float(x) for x in a[4:11]], [float(x) for x in b[4:11]], rtol=2e-3, atol=1e-9)
    # Keplerian, Jacobi centre: angles in degrees on both sides
    out = io.StringIO()
    logquery.run_with_args(args(keplerian="jacobi", time_range=Range.interval(0.0, 0.25), body_range=Range.interval(1, 3)), out)
    rows = [l.split() for l in out.getvalue().splitlines()]
    ref = cxx("-k", "-t", "0..0.25")
    assert len(rows) == len(ref) == 16 * 3 * 3
    for a, b in zip(rows, ref):
        assert a[:4] == b[:4]
        ea, eb = [float(x) for x in a[5:11]], [float(x) for x in b[5:11]]
        np.testing.assert_allclose(ea[:3], eb[:3], rtol=1e-4, atol=1e-6)          # a, e, i
        for x, y in zip(ea[3:], eb[3:]):                                           # angles modulo 360
            assert abs((x - y + 180) % 360 - 180) < 0.05 or eb[1] < 1e-3, (a, b)
    # body range, truncation, initial/final conditions, key listing
    out = io.StringIO()
    logquery.run_with_args(args(final_conditions=True, system_range=Range.interval(2, 3), body_range=Range.single(0)), out)
    rows = [l.split() for l in out.getvalue().splitlines()]
    assert [(int(r[2]), int(r[3])) for r in rows] == [(2, 0), (3, 0)] and all(abs(float(r[1]) - 1.0) < 0.011 for r in rows)
    out = io.StringIO()
    logquery.run_with_args(args(max_records=5), out)
    assert len(out.getvalue().splitlines()) == 5 * 4
    a = logquery.parse_cmd(["-d", fn, "-s", "1..2", "-t", "0.5", "-k", "astrocentric", "--initial-conditions"])
    assert a.system_range.ulPair() == (1, 2) and a.time_range.isSingle() and a.keplerian == "astrocentric"


def test_logrecord_helpers(oracle):
    from swarm_amd.logrecord import LogRecord, center_of_mass
    import swarm_amd.swarmng as swarmng
    raw = fixture_log(oracle).bytes()
    l = LogRecord.from_binary(raw[:288])
    assert l.msgid == 1 and len(l.bodies) == 4 and l.star() is l.bodies[0]
    assert LogRecord.from_binary(b"\xfe\xff\xff\xff" + raw[4:288]) is None          # a printf record
    (x, y, z), (vx, vy, vz), m = l.barycenter()
    assert abs(m - sum(b.mass for b in l.bodies)) < 1e-15
    assert abs(x - sum(b.position[0] * b.mass for b in l.bodies) / m) < 1e-15
    assert l.origin().mass == l.bodies[0].mass and l.origin().position == (0, 0, 0)
    assert center_of_mass(l.bodies[:1]) == (l.bodies[0].position, l.bodies[0].velocity, l.bodies[0].mass)
    orb = list(l.bodies_in_keplerian(l.star()))
    assert [i for i, _, _ in orb] == [1, 2, 3]
    for i, b, k in orb:
        assert abs(k.a - 1.4 ** (i - 1)) < 0.02 and k.e < 0.02 and k.i == 0.0
    # the swarmng-level names of the reference's tutorials
    k = swarmng.keplerian_for_cartesian(l.bodies[1], l.star())
    assert isinstance(k, swarmng.KeplerianCoordinates) and abs(k.a - orb[0][2].a) < 1e-15
    assert swarmng.logrecord.LogRecord is LogRecord and os.path.isdir(os.path.join(swarmng.SWARMDIR, "swarm_amd"))
    assert "jacobi" in swarmng.query.Keplerian.choices and swarmng.range_type.Range.single(1).contains(1)


@pytest.mark.gpu
def test_swarmng_init_routes_kernel_records_into_a_binary_log(gpu, tmp_path):
    """swarm::init(cfg) with log_writer=binary (swarm.h:76): integrators created afterwards log into the file; the
    Python-written file is byte-identical to the one the C++ CLI writes for the same run, and IndexedLogDB reads
    back 16 x 11 kernel-written snapshots that print as test/bdb/test.4.ref.txt through `swarm query`."""
    import swarm_amd.swarmng as swarmng
    fn = str(tmp_path / "py.log")
    cfg = swarmng.config(integrator="hermite_cpu_log", time_step=0.01, log_interval=0.1, nogpu=1,
                         log_writer="binary", log_output=fn)
    swarmng.init(cfg)
    try:
        ens = swarmng.DefaultEnsemble.load_from_text(os.path.join(GOLDEN, "bdb", "test.4.in.txt"))
        integ = swarmng.Integrator.create(cfg)
        integ.ensemble = ens
        integ.destination_time = 1.0
        integ.integrate()
    finally:
        swarmng.shutdown()
    d = swarmng.logdb.IndexedLogDB(fn)
    assert len(list(d.all_records())) == 176
    fc = dict(d.final_conditions(swarmng.range_type.Range.universal()))
    assert sorted(fc) == list(range(16))
    for s, l in fc.items():
        assert abs(l.time - 1.0) < 0.011 and l.sys == s
    p = subprocess.run([SWARM, "query", "-f", fn], capture_output=True, text=True, timeout=60)
    assert p.stdout == open(os.path.join(GOLDEN, "bdb", "test.4.ref.txt")).read()
    cfn = str(tmp_path / "cli.log")
    subprocess.run([SWARM, "integrate", "-I", os.path.join(GOLDEN, "bdb", "test.4.in.txt"), "integrator=hermite_cpu_log",
                    "time_step=0.01", "log_interval=0.1", "destination_time=1", "nogpu=1", "log_writer=binary",
                    "log_output=" + cfn], check=True, capture_output=True, timeout=120)
    assert open(cfn, "rb").read() == open(fn, "rb").read()
    # a second init() with the null writer detaches the file again
    swarmng.init(swarmng.config(nogpu=1))
    assert swarmng.Integrator.create(cfg)._impl._logman.capacity == 0
