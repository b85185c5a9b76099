"""Host and device ensembles.

HostEnsemble mirrors swarm::hostEnsemble's accessors over plain numpy AoS arrays;
DeviceEnsemble owns the wave-chunk device image (include/swarm_hip.h) -- the counterpart of
deviceEnsemble / EnsembleAlloc::cloneTo (swarm/ensemble_alloc.hpp:30-68)."""
import ctypes as C
import math
import weakref

import numpy as np

from . import capi

FLT_MAX = float(np.finfo(np.float32).max)


def _free_pinned(ptr):
    try:
        capi.lib.swarm_hip_host_free(C.c_void_p(ptr))
    except Exception:       # interpreter shutdown
        pass


def _pinned_block(nbytes):
    """page-locked host memory (hipHostMalloc) as a ctypes buffer; freed when the last numpy view of it dies"""
    p = C.c_void_p()
    capi.check(capi.lib.swarm_hip_host_alloc(C.byref(p), C.c_size_t(nbytes)))
    buf = (C.c_uint8 * nbytes).from_address(p.value)
    weakref.finalize(buf, _free_pinned, p.value)
    return buf


class HostEnsemble:
    """pinned=True is swarm::hostEnsemble (HostAllocator = cudaMallocHost, types/allocators.hpp:93-110,
    ensemble_alloc.hpp:66): the arrays live in one page-locked block, so upload/download run as DMA at PCIe
    speed; the default is defaultEnsemble (plain pageable memory)."""

    FIELDS = ("pos", "vel", "mass", "time", "body_attr", "sys_attr", "state", "id")

    def __init__(self, nsys, nbod, nbattr=1, nsattr=1, pinned=False):
        self.nsys, self.nbod, self.nbattr, self.nsattr = int(nsys), int(nbod), int(nbattr), int(nsattr)
        self.pinned = bool(pinned)
        shapes = {"pos": (nsys, nbod, 3), "vel": (nsys, nbod, 3), "mass": (nsys, nbod), "time": (nsys,),
                  "body_attr": (nsys, nbod, nbattr), "sys_attr": (nsys, nsattr), "state": (nsys,), "id": (nsys,)}
        if pinned:
            sizes = {k: int(np.prod(shapes[k])) * (4 if k in ("state", "id") else 8) for k in self.FIELDS}
            buf = _pinned_block(sum(sizes.values()) + 64)
            off = 0
            for k in self.FIELDS:       # doubles first, so every array is 8-byte aligned
                dt = np.int32 if k in ("state", "id") else np.float64
                a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shapes[k])), offset=off).reshape(shapes[k])
                a[...] = 0
                setattr(self, k, a)
                off += sizes[k]
        else:
            for k in self.FIELDS:
                setattr(self, k, np.zeros(shapes[k], dtype=np.int32 if k in ("state", "id") else np.float64))
        self.id[:] = np.arange(nsys, dtype=np.int32)

    @staticmethod
    def create(nbod, nsys):
        return HostEnsemble(nsys, nbod)

    def clone(self, pinned=None):
        pinned = self.pinned if pinned is None else pinned
        e = HostEnsemble(self.nsys, self.nbod, self.nbattr, self.nsattr, pinned=pinned)
        for k in self.FIELDS:
            if pinned:
                getattr(e, k)[...] = getattr(self, k)
            else:
                setattr(e, k, getattr(self, k).copy())
        return e

    def slice(self, lo, hi):
        e = HostEnsemble(hi - lo, self.nbod, self.nbattr, self.nsattr)
        for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
            setattr(e, k, getattr(self, k)[lo:hi].copy())
        return e

    # swarm/types/ensemble.hpp:402-413 (host twin; the device version is swarm_hip_total_energy)
    def calc_total_energy(self):
        K = (0.5 * self.mass * (self.vel ** 2).sum(-1)).sum(-1)
        U = np.zeros(self.nsys)
        for i in range(self.nbod):
            for j in range(i):
                d = np.sqrt(((self.pos[:, i] - self.pos[:, j]) ** 2).sum(-1))
                U += -self.mass[:, i] * self.mass[:, j] / d
        return K + U


def generate_ensemble(cfg):
    """swarm/utils.cpp:63-103 (deterministic in the reference: theta = bod*pi/4)."""
    nsys, nbod = cfg.require("nsys", 0), cfg.require("nbod", 0)
    spacing_factor = cfg.optional("spacing_factor", 1.4)
    planet_mass = cfg.optional("planet_mass", .0001)
    ejection_factor = cfg.optional("ejection_factor", 1.0 / math.sqrt(2))
    star_radius = cfg.optional("star_radius", 0.1)
    planet_radius = cfg.optional("planet_radius", 0.001)
    e = HostEnsemble(nsys, nbod)
    e.mass[:, 0] = 1.
    e.body_attr[:, 0, 0] = star_radius
    for bod in range(1, nbod):
        rmag = 10 * math.pow(spacing_factor, int(bod - 1))
        vmag = math.sqrt(2 * 1. / rmag) * ejection_factor
        theta = bod * math.pi / 4
        e.pos[:, bod] = (rmag * math.cos(theta), rmag * math.sin(theta), 0)
        e.vel[:, bod] = (-vmag * math.sin(theta), vmag * math.cos(theta), 0.)
        e.mass[:, bod] = planet_mass
        e.body_attr[:, bod, 0] = planet_radius
    e.state[:] = capi.ACTIVE
    e.time[:] = 0
    e.id[:] = np.arange(nsys)
    return e


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceBuffer:
    """hipMalloc'ed bytes owned by this object."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        capi.check(capi.lib.swarm_hip_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value or 0

    def zero(self, stream=None):
        capi.check(capi.lib.swarm_hip_memset(self.ptr, 0, self.nbytes, stream))

    def to_host(self, dtype=np.uint8, nbytes=None):
        n = self.nbytes if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        capi.check(capi.lib.swarm_hip_memcpy_d2h(_ptr(out), self.ptr, n))
        return out.view(dtype)

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        capi.check(capi.lib.swarm_hip_memcpy_h2d(self.ptr, _ptr(arr), arr.nbytes))

    def free(self):
        if self.ptr:
            capi.lib.swarm_hip_free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceEnsemble:
    """Wave-chunk device image.  Buffers are either owned (hipMalloc) or borrowed device
    pointers (e.g. torch tensors in bench.py)."""

    def __init__(self, nsys, nbod, nbattr=1, nsattr=1, pointers=None):
        self.nsys, self.nbod, self.nbattr, self.nsattr = int(nsys), int(nbod), int(nbattr), int(nsattr)
        L = capi.lib
        self.body_bytes = L.swarm_hip_body_bytes(nsys, nbod, nbattr)
        self.sysd_bytes = L.swarm_hip_sysd_bytes(nsys, nsattr)
        self.sysi_bytes = L.swarm_hip_sysi_bytes(nsys)
        if pointers is None:
            self._own = [DeviceBuffer(max(self.body_bytes, 8)), DeviceBuffer(max(self.sysd_bytes, 8)),
                         DeviceBuffer(max(self.sysi_bytes, 8))]
            pointers = [b.ptr for b in self._own]
        else:
            self._own = []
        self.c = capi.EnsembleStruct(self.nsys, self.nbod, self.nbattr, self.nsattr, *pointers)
        # ACTIVE systems as last counted (count_active), None when something may have changed the states since:
        # GpuIntegrator hands it to the next launch as swarm_hip_launch_t::active_hint
        self.last_active_count = None

    @property
    def ref(self):
        return C.byref(self.c)

    def chunk_view(self, c0, c1):
        """The systems of wave chunks [c0, c1) as an ensemble of their own on borrowed pointers: a run of chunks is
        one contiguous byte range of each of the three buffers (include/swarm_hip.h), which is also what a multi-GPU
        shard is (parallel.py).  System ids travel with the systems (sysi)."""
        nchunk = int(capi.lib.swarm_hip_nchunk(self.nsys))
        assert 0 <= c0 < c1 <= nchunk, (c0, c1, nchunk)
        nsys = min(self.nsys, c1 * capi.CHUNK) - c0 * capi.CHUNK
        W = capi.CHUNK
        off = (c0 * self.nbod * (7 + self.nbattr) * W * 8, c0 * (1 + self.nsattr) * W * 8, c0 * 2 * W * 4)
        v = DeviceEnsemble(nsys, self.nbod, self.nbattr, self.nsattr,
                           pointers=[self.c.body + off[0], self.c.sysd + off[1], self.c.sysi + off[2]])
        v.first_system = c0 * W
        return v

    def upload(self, h):
        assert (h.nsys, h.nbod) == (self.nsys, self.nbod)
        arrs = [np.ascontiguousarray(a) for a in (h.pos, h.vel, h.mass, h.time, h.state, h.id,
                                                  h.body_attr, h.sys_attr)]
        assert arrs[4].dtype == np.int32 and arrs[5].dtype == np.int32
        capi.check(capi.lib.swarm_hip_upload_aos(self.ref, *[_ptr(a) for a in arrs]))
        self.last_active_count = None

    def download(self, h=None):
        h = h if h is not None else HostEnsemble(self.nsys, self.nbod, self.nbattr, self.nsattr)
        capi.check(capi.lib.swarm_hip_download_aos(self.ref, _ptr(h.pos), _ptr(h.vel), _ptr(h.mass),
                                                   _ptr(h.time), _ptr(h.state), _ptr(h.id),
                                                   _ptr(h.body_attr), _ptr(h.sys_attr)))
        return h

    def count_active(self, stream=None):
        n = C.c_int32()
        capi.check(capi.lib.swarm_hip_count_active(self.ref, stream, C.byref(n)))
        self.last_active_count = n.value
        return n.value

    def reactivate(self, stream=None):
        capi.check(capi.lib.swarm_hip_reactivate(self.ref, stream))
        self.last_active_count = None

    def total_energy(self, stream=None):
        n = capi.lib.swarm_hip_nchunk(self.nsys) * capi.CHUNK
        buf = DeviceBuffer(max(8 * n, 8))
        capi.check(capi.lib.swarm_hip_total_energy(self.ref, buf.ptr, stream))
        capi.check(capi.lib.swarm_hip_synchronize(stream))
        return buf.to_host(np.float64)[: self.nsys].copy()
