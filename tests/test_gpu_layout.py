"""GPU checks of the data-layout entry points: AoS upload/download, the reference
CoalescedStructArray image <-> wave chunks (bit-exact byte work), active count / reactivate,
energy and the statistics reduction."""
import ctypes as C

import numpy as np
import pytest

from conftest import perturbed, to_host

pytestmark = pytest.mark.gpu


def ref_image(e, RC, nbattr=1, nsattr=1):
    """numpy restatement of the reference memory image: ensemble.hpp:131-150,227-238,499-517"""
    nsys, nbod = e.nsys, e.nbod
    nblk = (nsys + RC - 1) // RC
    body = np.zeros((nblk, nbod, 7 + nbattr, RC))
    sysb = np.zeros((nblk, 2 + nsattr, RC))
    bunch = np.zeros((nblk, RC, 2), dtype=np.int32)
    for i in range(nsys):
        g, l = divmod(i, RC)
        for c in range(3):
            body[g, :, 2 * c, l] = e.pos[i, :, c]
            body[g, :, 2 * c + 1, l] = e.vel[i, :, c]
        body[g, :, 6, l] = e.mass[i]
        body[g, :, 7:, l] = e.body_attr[i]
        sysb[g, 0, l] = e.time[i]
        bunch[g, l] = (e.state[i], e.id[i])
        sysb[g, 2:, l] = e.sys_attr[i]
    sysb[:, 1, :] = bunch.reshape(nblk, RC * 2).view(np.float64)
    return body, sysb


@pytest.mark.parametrize("nsys,nbod", [(1, 3), (64, 3), (65, 4), (1000, 6), (4099, 8), (130, 10)])
def test_upload_download_roundtrip(gpu, oracle, nsys, nbod):
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=2)
    e.state[:] = np.random.default_rng(0).integers(-1, 3, nsys)
    e.sys_attr[:] = np.arange(nsys)[:, None] * 0.5
    h = to_host(S, e)
    d = S.DeviceEnsemble(nsys, nbod)
    d.upload(h)
    back = d.download()
    for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
        assert (getattr(back, k) == getattr(h, k)).all(), k
    assert d.count_active() == int((e.state == 0).sum())
    d.reactivate()
    back = d.download()
    want = e.state.copy()
    want[want == 1] = 0
    assert (back.state == want).all()
    en = d.total_energy()
    np.testing.assert_array_equal(en, O.total_energy(e))      # same operation order, no FMA: bit-exact


@pytest.mark.parametrize("nbod,nbattr,nsattr", [(3, 1, 1), (5, 3, 1), (4, 4, 2), (8, 3, 2), (10, 13, 1)])
def test_upload_download_roundtrip_with_attributes(gpu, nbod, nbattr, nsattr):
    """AoS <-> wave chunks with several body and system attributes (the LDS-tiled kernel's body-attribute group has
    F = nbod * nbattr columns), a ragged last chunk, and NULL attribute arrays on upload (zero-filled on the device).
    nbod = 10 with 13 attributes needs a 67 KB tile, more than the tiled kernel takes: that case goes through the
    lane-per-system conversion kernel."""
    S = gpu
    nsys = 64 * 3 + 17
    rng = np.random.default_rng(nbod * 100 + nbattr)
    h = S.HostEnsemble(nsys, nbod, nbattr, nsattr)
    for k in ("pos", "vel", "mass", "time", "body_attr", "sys_attr"):
        getattr(h, k)[...] = rng.standard_normal(getattr(h, k).shape)
    h.state[:] = rng.integers(-1, 3, nsys)
    h.id[:] = rng.permutation(nsys)
    d = S.DeviceEnsemble(nsys, nbod, nbattr, nsattr)
    d.upload(h)
    back = d.download()
    for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
        assert (getattr(back, k) == getattr(h, k)).all(), k
    # the device image itself: row (b, 7 + a) of chunk k holds attribute a of body b for the chunk's 64 systems
    img = S.DeviceBuffer.to_host(d._own[0], np.float64).reshape(-1, nbod, 7 + nbattr, 64)
    k, l = np.arange(nsys) // 64, np.arange(nsys) % 64
    for a in range(nbattr):
        assert (img[k, nbod - 1, 7 + a, l] == h.body_attr[:, nbod - 1, a]).all()
    assert (img[-1, :, :, 17:] == 0).all()                       # padding lanes of the ragged chunk
    # NULL attribute arrays: zeros on the device, everything else as given
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    S.capi.check(S.capi.lib.swarm_hip_upload_aos(d.ref, ptr(h.pos), ptr(h.vel), ptr(h.mass), ptr(h.time), ptr(h.state),
                                                 ptr(h.id), None, None))
    back = d.download()
    assert (back.body_attr == 0).all() and (back.sys_attr == 0).all()
    for k in ("pos", "vel", "mass", "time", "state", "id"):
        assert (getattr(back, k) == getattr(h, k)).all(), k


def test_pinned_host_ensemble_and_kept_staging_buffer(gpu, oracle):
    """hostEnsemble (page-locked, types/allocators.hpp:93-110): same bytes through upload / integrate / download as
    the pageable defaultEnsemble; its arrays outlive the ensemble object.  The calls go large -> small -> large so
    that the device's kept staging buffer is reused at a smaller size and then regrown."""
    S, O = gpu, oracle
    outs = []
    for nsys, nbod in [(3000, 6), (70, 3), (5000, 8)]:
        e = perturbed(O, nsys, nbod, seed=7)
        e.sys_attr[:] = np.arange(nsys)[:, None] * 0.25
        plain = to_host(S, e)
        pinned = plain.clone(pinned=True)
        assert pinned.pinned and not plain.pinned
        for k in S.HostEnsemble.FIELDS:
            assert (getattr(pinned, k) == getattr(plain, k)).all(), k
        for h in (plain, pinned):
            integ = S.Integrator.create(S.config({"integrator": "hermite", "time_step": "1e-3"}))
            integ.set_ensemble(h)
            integ.set_destination_time(0.05)
            integ.integrate()
        for k in S.HostEnsemble.FIELDS:
            assert (getattr(pinned, k) == getattr(plain, k)).all(), k
        assert (pinned.time == 0.05).all()
        outs.append((pinned.pos, plain.pos.copy()))
        del pinned
    import gc
    gc.collect()
    for a, b in outs:       # views keep the page-locked block alive
        assert (a == b).all()


@pytest.mark.parametrize("RC", [4, 8, 16, 64])
@pytest.mark.parametrize("nsys,nbod", [(8, 3), (100, 3), (777, 5), (2048, 8)])
def test_reference_layout_import_export(gpu, oracle, RC, nsys, nbod):
    S, O = gpu, oracle
    e = perturbed(O, nsys, nbod, seed=4)
    e.state[:] = np.random.default_rng(1).integers(-1, 3, nsys)
    e.time[:] = np.linspace(0, 1, nsys)
    e.sys_attr[:, 0] = np.arange(nsys)
    h = to_host(S, e)
    body, sysb = ref_image(e, RC)
    db, ds = S.DeviceBuffer(body.nbytes), S.DeviceBuffer(sysb.nbytes)
    db.from_host(body)
    ds.from_host(sysb)
    d = S.DeviceEnsemble(nsys, nbod)
    S.capi.check(S.capi.lib.swarm_hip_import_ref_layout(d.ref, db.ptr, ds.ptr, RC, None))
    got = d.download()
    for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
        assert (getattr(got, k) == getattr(h, k)).all(), k
    # export into zeroed buffers reproduces the image bit for bit
    db2, ds2 = S.DeviceBuffer(body.nbytes), S.DeviceBuffer(sysb.nbytes)
    db2.zero()
    ds2.zero()
    S.capi.check(S.capi.lib.swarm_hip_export_ref_layout(d.ref, db2.ptr, ds2.ptr, RC, None))
    S.capi.check(S.capi.lib.swarm_hip_synchronize(None))
    assert db2.to_host(np.float64).tobytes() == body.tobytes()
    assert ds2.to_host(np.float64).tobytes() == sysb.tobytes()
    with pytest.raises(S.capi.SwarmHipError):
        S.capi.check(S.capi.lib.swarm_hip_import_ref_layout(d.ref, db.ptr, ds.ptr, 7, None))


def test_energy_stats_reduction(gpu, oracle):
    S, O = gpu, oracle
    nsys = 1000
    e = perturbed(O, nsys, 3, seed=6)
    e.state[:] = np.random.default_rng(2).integers(-1, 3, nsys)
    h = to_host(S, e)
    d = S.DeviceEnsemble(nsys, 3)
    d.upload(h)
    e0 = O.total_energy(e)
    e1 = e0 * (1 + 1e-9 * np.random.default_rng(3).standard_normal(nsys))
    b0, b1, out = S.DeviceBuffer(8 * nsys), S.DeviceBuffer(8 * nsys), S.DeviceBuffer(64)
    b0.from_host(e0)
    b1.from_host(e1)
    S.capi.check(S.capi.lib.swarm_hip_energy_stats(d.ref, b0.ptr, b1.ptr, out.ptr, None))
    S.capi.check(S.capi.lib.swarm_hip_synchronize(None))
    r = out.to_host(np.float64)
    err = np.abs((e1 - e0) / e0)
    np.testing.assert_allclose(r[0], err.sum(), rtol=1e-12)
    assert r[1] == err.max() and r[2] == err.min()
    assert [int(x) for x in r[3:8]] == [int((e.state == 0).sum()), int((e.state == 1).sum()),
                                        int((e.state == -1).sum()), int((e.state == 2).sum()), nsys]
