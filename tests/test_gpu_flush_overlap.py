"""The two-log manager (LogManager(..., buffers=2)): the copy of the log one launch filled runs on a second
stream while the next launch -- the next batch of wave chunks of the same attempt (`log_batches`), or the next attempt
of the launch loop -- appends to the other log.  Against the reference's protocol (one log, one synchronous copy after
every launch, swarm/log/logmanager.cpp:67-87) the writer must receive the same records -- every one of them, byte for
byte, none dropped, none twice; their order in the stream is free in both -- and the ensembles must be bit-identical."""
import numpy as np
import pytest

from conftest import perturbed, to_host

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "vel", "time", "state", "id", "mass")


def split(raw):
    out, off = [], 0
    while off < len(raw):
        ln = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=off + 4)[0])
        out.append(raw[off:off + ln])
        off += ln
    return sorted(out)


def run(S, e, cfg, dest, buffers, cap, stream=None):
    h = to_host(S, e)
    integ = S.Integrator.create(S.config(cfg))
    integ.set_ensemble(h)
    integ.set_destination_time(dest)
    integ.stream = stream
    lm = S.LogManager(cap, "host", buffers=buffers)
    lm.replay_printf = False
    integ.set_log_manager(lm)
    attempts = integ.integrate()
    return h, bytes(lm.host), lm.dropped, attempts, lm


ADAP = {"integrator": "hermite_adap_log", "time_step_factor": "0.017", "min_time_step": "1e-5", "log_interval": "0.5"}


@pytest.mark.parametrize("nbod,nsys,batches", [(4, 8192, 4), (4, 1000, 3), (3, 130, 16), (6, 4096, 2)])
def test_batches_of_one_attempt_give_the_same_records(gpu, oracle, nbod, nsys, batches):
    """one attempt as `batches` launches over runs of wave chunks (more batches than chunks: one chunk each)"""
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=40 + nbod)
    e.time[:] = np.linspace(0.0, 0.05, nsys)
    e.state[5::13] = O.DISABLED
    e.id[:] = 7000 + np.arange(nsys)
    rec = S.capi.lib.swarm_hip_snapshot_record_bytes(nbod)
    cap = 64 * rec * nsys     # room for every record of the run in ONE log, so that nothing is dropped either way
    h1, raw1, d1, a1, _ = run(S, e, ADAP, 10.0, 1, cap)
    h2, raw2, d2, a2, lm = run(S, e, dict(ADAP, log_batches=str(batches)), 10.0, 2, cap)
    assert a1 == a2 == 1 and d1 == d2 == 0
    r1, r2 = split(raw1), split(raw2)
    n_active = int((e.state == O.ACTIVE).sum())
    assert 20 * n_active <= len(r1) <= 21 * n_active     # t0 and every 0.5 up to 10 (the t0 record: hermite_adap.hpp:130)
    assert r1 == r2
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all(), k
    assert lm.bytes_flushed == len(raw1) and lm._pending == [None, None]


def test_copies_overlap_the_launches_of_later_attempts_and_lose_nothing(gpu, oracle):
    """a launch loop of many attempts (max_iterations caps each launch; log_time_interval is armed again at every
    launch and logs the state it starts from, log_time_interval.hpp:118-119): every flush but the last overlaps the
    next attempt's kernel, which appends to the other log"""
    S, O = gpu, oracle
    nsys, nbod = 3000, 4
    e = perturbed(O, nsys, nbod, seed=77)
    e.id[:] = np.arange(nsys)
    cfg = dict(ADAP, max_iterations="5")
    cap = 8 << 20       # two or three attempts' worth: the logs are reused many times
    h1, raw1, d1, a1, _ = run(S, e, cfg, 30.0, 1, cap)
    h2, raw2, d2, a2, _ = run(S, e, cfg, 30.0, 2, cap)
    h3, raw3, d3, a3, _ = run(S, e, dict(cfg, log_batches="5"), 30.0, 2, cap)
    assert a1 == a2 == a3 and a1 > 10 and d1 == d2 == d3 == 0
    r1 = split(raw1)
    assert len(r1) >= 61 * nsys and r1 == split(raw2) == split(raw3)
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all() and (getattr(h1, k) == getattr(h3, k)).all(), k


def test_overflow_is_counted_not_silent(gpu, oracle):
    """a log too small for a launch's records: both managers drop the surplus and count it (the reference drops
    silently, gpulog_write.h:14-22); records written + records dropped = records due"""
    S, O = gpu, oracle
    nsys, nbod = 2048, 3
    e = perturbed(O, nsys, nbod, seed=5)
    rec = S.capi.lib.swarm_hip_snapshot_record_bytes(nbod)
    cap = rec * nsys * 3
    for buffers, extra in ((1, {}), (2, {}), (2, {"log_batches": "2"})):
        h, raw, dropped, attempts, _ = run(S, e, dict(ADAP, **extra), 10.0, buffers, cap)
        got = len(split(raw))
        assert got + dropped == 21 * nsys and dropped > 0, (buffers, extra, got, dropped)


def test_launches_on_a_stream_of_their_own(gpu, oracle):
    """the launch stream need not be the NULL stream: kernels, the cursor copy and the clearing of a reused log's cursor
    are ordered by the caller's stream, the copies by the manager's"""
    import ctypes as C
    S, O = gpu, oracle
    nsys, nbod = 2000, 4
    e = perturbed(O, nsys, nbod, seed=9)
    cfg = dict(ADAP, max_iterations="6")
    cap = 8 << 20
    stream = C.c_void_p()
    S.capi.check(S.capi.lib.swarm_hip_stream_create(C.byref(stream)))
    try:
        h1, raw1, d1, a1, _ = run(S, e, cfg, 20.0, 1, cap)
        h2, raw2, d2, a2, _ = run(S, e, dict(cfg, log_batches="3"), 20.0, 2, cap, stream=stream)
        h3, raw3, d3, a3, _ = run(S, e, cfg, 20.0, 1, cap, stream=stream)
    finally:
        S.capi.check(S.capi.lib.swarm_hip_stream_destroy(stream))
    assert a1 == a2 == a3 > 5 and d1 == d2 == d3 == 0
    assert split(raw1) == split(raw2) == split(raw3) and len(raw1) > 40 * 288 * nsys
    for k in FIELDS:
        assert (getattr(h1, k) == getattr(h2, k)).all() and (getattr(h1, k) == getattr(h3, k)).all(), k
