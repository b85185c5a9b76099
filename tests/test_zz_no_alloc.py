"""swarm_hip_count_active_noalloc (include/swarm_hip.h): the allocation-free variant of
swarm_hip_count_active, checked against that entry point through the C-ABI -- bit-exact (an
integer count) -- on two calls in a row, the second with a smaller ensemble, so that the static
partial-count array still holds the first call's contents when the second one runs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ensemble(S, nsys, nbod, seed):
    h = S.HostEnsemble(nsys, nbod)
    rng = np.random.default_rng(seed)
    h.mass[:] = 1.0
    h.pos[:] = rng.standard_normal(h.pos.shape)
    # every state value the library knows, ACTIVE in about a third of the systems
    h.state[:] = rng.choice(np.array([0, 0, 1, 2, -1, 1], dtype=np.int32), size=nsys)
    d = S.DeviceEnsemble(nsys, nbod)
    d.upload(h)
    return h, d


def _count_noalloc(S, d, d_count):
    S.capi.check(S.capi.lib.swarm_hip_count_active_noalloc(d.ref, d_count.ptr, None))
    S.capi.check(S.capi.lib.swarm_hip_synchronize(None))
    return int(d_count.to_host(np.int32)[0])


def test_noalloc_count_equals_count_active_twice_second_smaller(gpu):
    S = gpu
    d_count = S.DeviceBuffer(16)
    d_count.from_host(np.full(4, 0x7FFFFFFF, dtype=np.int32))   # the entry point writes, it does not accumulate
    # 3 workgroups and a ragged last chunk, then a smaller ensemble inside one workgroup
    for nsys, nbod, seed in ((3 * 4096 + 777, 3, 1), (1000, 4, 2)):
        h, d = _ensemble(S, nsys, nbod, seed)
        want = d.count_active()
        assert want == int((h.state == 0).sum()) and 0 < want < nsys
        assert _count_noalloc(S, d, d_count) == want
    # an ensemble with no ACTIVE system gives 0, not the last call's value
    h, d = _ensemble(S, 130, 3, 3)
    h.state[:] = 1
    d.upload(h)
    assert d.count_active() == 0 and _count_noalloc(S, d, d_count) == 0


def test_noalloc_rejects_what_it_cannot_hold(gpu):
    S = gpu
    big = S.capi.EnsembleStruct(S.capi.NOALLOC_MAX_NSYS + 1, 3, 1, 1, 8, 8, 8)   # never dereferenced: rejected first
    d_count = S.DeviceBuffer(16)
    rc = S.capi.lib.swarm_hip_count_active_noalloc(C.byref(big), d_count.ptr, None)
    assert rc == -1 and b"SWARM_HIP_NOALLOC_MAX_NSYS" in S.capi.lib.swarm_hip_last_error()
    _, d = _ensemble(S, 64, 3, 4)
    assert S.capi.lib.swarm_hip_count_active_noalloc(d.ref, None, None) == -1


def test_noalloc_count_on_a_stream_of_its_own(gpu):
    """the call is asynchronous on the caller's stream: two calls queued back to back on a non-NULL stream (ordered
    by the stream, as the header asks of calls that share a GPU), the results read after one synchronisation"""
    S = gpu
    L = S.capi.lib
    stream = C.c_void_p()
    S.capi.check(L.swarm_hip_stream_create(C.byref(stream)))
    try:
        (h1, d1), (h2, d2) = _ensemble(S, 2 * 4096 + 5, 3, 11), _ensemble(S, 700, 5, 12)
        L.swarm_hip_synchronize(None)
        c1, c2 = S.DeviceBuffer(16), S.DeviceBuffer(16)
        S.capi.check(L.swarm_hip_count_active_noalloc(d1.ref, c1.ptr, stream))
        S.capi.check(L.swarm_hip_count_active_noalloc(d2.ref, c2.ptr, stream))
        S.capi.check(L.swarm_hip_synchronize(stream))
        assert int(c1.to_host(np.int32)[0]) == int((h1.state == 0).sum())
        assert int(c2.to_host(np.int32)[0]) == int((h2.state == 0).sum())
    finally:
        S.capi.check(L.swarm_hip_stream_destroy(stream))
