"""`swarmng`-compatible Python surface (py/swarmng/__init__.py + src/python/module.cpp:181-294 of
the reference, which needs boost.python and Python 2): the same names and call shapes over this
package's ctypes binding, so scripts like py/tests/trivial.py run with `import swarm_amd.swarmng as
swarmng`.  Ensembles are views onto HostEnsemble's numpy arrays."""
import math
import os

import numpy as np

from . import capi
from . import logdb, logrecord, range_type
from . import logquery as query
from .logrecord import KeplerianCoordinates, keplerian_for_cartesian
from . import snapshot as _snap
from .config import config as _config
from .ensemble import HostEnsemble
from .ensemble import generate_ensemble as _generate
from .integrator import GpuIntegrator as _GpuIntegrator
from .integrator import Integrator as _Integrator
from .integrator import LogManager


SWARMDIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))   # py/swarmng/__init__.py:22


_default_log = None   # log::manager::default_log(), logmanager.cpp:37-47


def _close_default_log():
    global _default_log
    if _default_log is not None:
        _default_log.close()
        _default_log = None


def init(cfg):
    """swarm::init(cfg), swarm/swarm.h:49-77: select the device, then initialise the default log from
    `log_writer` (null | host_array | binary + `log_output`); every integrator created afterwards logs there
    (integrator.cpp:51).  The binary writer's file is sorted and finished when the process exits, when init()
    is called again, or on shutdown()."""
    global _default_log
    capi.check(capi.lib.swarm_hip_init(cfg.optional("HIP_DEVICE", cfg.optional("CUDA_DEVICE", 0))))
    _close_default_log()
    if cfg.optional("log_writer", "null") != "null":
        import atexit
        _default_log = LogManager.from_config(cfg)
        atexit.register(_close_default_log)


def shutdown():
    """log::manager::shutdown(), logmanager.cpp:59-64: back to the null writer"""
    _close_default_log()


def sync():
    capi.check(capi.lib.swarm_hip_synchronize(None))


class Config(_config):
    @staticmethod
    def load(path, initial=None):
        return Config(_config.load(path, initial))


def config(*l, **kw):
    """swarmng.config(dict..., key=value...) -> Config with stringified values (__init__.py:77-105)"""
    c = Config()
    for h in list(l) + [kw]:
        for k, v in h.items():
            c[str(k)] = str(v)
    return c


mkConfig = config


class _Attributes:
    def __init__(self, arr):
        self._a = arr

    def __len__(self):
        return len(self._a)

    def __getitem__(self, i):
        return float(self._a[i])

    def __setitem__(self, i, v):
        self._a[i] = v


class _Component:
    def __init__(self, e, i, b, c):
        self._e, self._k = e, (i, b, c)

    pos = property(lambda s: float(s._e.pos[s._k]), lambda s, v: s._e.pos.__setitem__(s._k, v))
    vel = property(lambda s: float(s._e.vel[s._k]), lambda s, v: s._e.vel.__setitem__(s._k, v))


class Body:
    def __init__(self, e, i, b):
        self._e, self._i, self._b = e, i, b

    mass = property(lambda s: float(s._e.mass[s._i, s._b]), lambda s, v: s._e.mass.__setitem__((s._i, s._b), v))
    pos = property(lambda s: [float(x) for x in s._e.pos[s._i, s._b]],
                   lambda s, v: s._e.pos.__setitem__((s._i, s._b), v))
    vel = property(lambda s: [float(x) for x in s._e.vel[s._i, s._b]],
                   lambda s, v: s._e.vel.__setitem__((s._i, s._b), v))
    attributes = property(lambda s: _Attributes(s._e.body_attr[s._i, s._b]))

    def distance_to_origin(self):
        return math.sqrt(float((self._e.pos[self._i, self._b] ** 2).sum()))

    def speed(self):
        return math.sqrt(float((self._e.vel[self._i, self._b] ** 2).sum()))

    def __getitem__(self, c):
        if not 0 <= c < 3:
            raise IndexError("")
        return _Component(self._e, self._i, self._b, c)


class System:
    def __init__(self, e, i):
        self._e, self._i = e, i

    def __len__(self):
        return self._e.nbod

    def __getitem__(self, b):
        if not 0 <= b < self._e.nbod:
            raise IndexError("")
        return Body(self._e, self._i, b)

    time = property(lambda s: float(s._e.time[s._i]), lambda s, v: s._e.time.__setitem__(s._i, v))
    id = property(lambda s: int(s._e.id[s._i]), lambda s, v: s._e.id.__setitem__(s._i, v))
    state = property(lambda s: int(s._e.state[s._i]), lambda s, v: s._e.state.__setitem__(s._i, v))
    attributes = property(lambda s: _Attributes(s._e.sys_attr[s._i]))
    total_energy = property(lambda s: float(s._e.slice(s._i, s._i + 1).calc_total_energy()[0]))

    def is_active(self):
        return self.state == capi.ACTIVE

    def is_inactive(self):
        return self.state == capi.INACTIVE

    def is_enabled(self):
        return self.state != capi.DISABLED

    def set_active(self):
        self.state = capi.ACTIVE

    def set_inactive(self):
        self.state = capi.INACTIVE

    def set_disabled(self):
        self.state = capi.DISABLED


class DefaultEnsemble(HostEnsemble):
    """swarm::defaultEnsemble as exported to Python (module.cpp:246-257)"""

    @staticmethod
    def create(nbod, nsys):
        return DefaultEnsemble(nsys, nbod)

    @staticmethod
    def _wrap(h):
        e = DefaultEnsemble(h.nsys, h.nbod, h.nbattr, h.nsattr)
        for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
            setattr(e, k, getattr(h, k))
        return e

    def clone(self):
        return DefaultEnsemble._wrap(HostEnsemble.clone(self))

    def __len__(self):
        return self.nsys

    def __getitem__(self, i):
        if not 0 <= i < self.nsys:
            raise IndexError("")
        return System(self, i)

    def save_to_text(self, fn):
        _snap.save_text(self, fn)

    def save_to_bin(self, fn):
        _snap.save(self, fn)

    @staticmethod
    def load_from_text(fn):
        return DefaultEnsemble._wrap(_snap.load_text(fn))

    @staticmethod
    def load_from_bin(fn):
        return DefaultEnsemble._wrap(_snap.load(fn))


Ensemble = DefaultEnsemble


def generate_ensemble(cfg):
    return DefaultEnsemble._wrap(_generate(cfg))


class Integrator:
    """swarmng.Integrator: create(cfg), .ensemble, .destination_time, .integrate()"""

    def __init__(self, impl):
        self._impl = impl

    @staticmethod
    def create(cfg):
        return GpuIntegrator._with_default_log(_Integrator.create(cfg))

    def integrate(self):
        self._impl.integrate()

    def _get_ens(self):
        return self._impl.get_ensemble()

    def _set_ens(self, e):
        self._impl.set_ensemble(e)

    ensemble = property(_get_ens, _set_ens)
    destination_time = property(lambda s: s._impl.get_destination_time(),
                                lambda s, t: s._impl.set_destination_time(t))


class GpuIntegrator(Integrator):
    @staticmethod
    def _with_default_log(impl):
        if _default_log is not None:
            impl.set_log_manager(_default_log)
        return GpuIntegrator(impl)

    @staticmethod
    def create(cfg):
        return GpuIntegrator._with_default_log(_Integrator.create(cfg))

    def core_integrate(self):
        return self._impl.core_integrate()

    def upload_ensemble(self):
        self._impl.upload_ensemble()

    def download_ensemble(self):
        self._impl.download_ensemble()

    def set_log_manager(self, lm):
        self._impl.set_log_manager(lm)


def calc_keplerian_for_cartesian(x, y, z, vx, vy, vz, GM):
    """swarmng.calc_keplerian_for_cartesian (src/python/module.cpp:65-73,186): -> [a, e, i, O, w, M], angles in
    radians; conventions of tutorials/kepler.cpp:102-200 (through libswarmng_hip.so)"""
    import ctypes as C
    import os
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libswarmng_hip.so"))
    f = lib.swarmng_calc_keplerian_for_cartesian
    f.argtypes = [C.POINTER(C.c_double)] + [C.c_double] * 7
    out = (C.c_double * 6)()
    f(out, x, y, z, vx, vy, vz, GM)
    return list(out)


def calc_cartesian_for_keplerian(a, e, i, O, w, M, GM=1.0):
    """swarmng.calc_cartesian_for_keplerian (module.cpp:79-91,187) = calc_cartesian_for_ellipse
    (tutorials/kepler.cpp:30-101): elliptic elements -> [x, y, z, vx, vy, vz].  Kepler's equation by the
    cube-root start and three third-order corrections of Mercury's orbel_ehie, as there.  The reference's
    wrapper hands an uninitialised GM to the C++ routine; here GM is an argument (default 1)."""
    import math

    def improve(e, M, x):
        sx, cx = math.sin(x), math.cos(x)
        es, ec = e * sx, e * cx
        f, fp, fpp, fppp = x - es - M, 1. - ec, es, ec
        dx = -f / fp
        dx = -f / (fp + dx * fpp / 2.)
        dx = -f / (fp + dx * fpp / 2. + dx * dx * fppp / 6.)
        return x + dx

    M = M - int(M / (2. * math.pi)) * 2. * math.pi
    if M < 0.:
        M += 2. * math.pi
    flip = M > math.pi
    if flip:
        M = 2. * math.pi - M
    x = (6. * M) ** (1. / 3.)
    for _ in range(3):
        x = improve(e, M, x)
    cape = 2. * math.pi - x if flip else x
    scap, ccap = math.sin(cape), math.cos(cape)
    sqe, sqgma = math.sqrt(1. - e * e), math.sqrt(GM * a)
    xfac1, xfac2 = a * (ccap - e), a * sqe * scap
    ri = 1. / (a * (1. - e * ccap))
    vfac1, vfac2 = -ri * sqgma * scap, ri * sqgma * sqe * ccap
    sw, cw, so, co, si, ci = math.sin(w), math.cos(w), math.sin(O), math.cos(O), math.sin(i), math.cos(i)
    d1 = (cw * co - sw * so * ci, cw * so + sw * co * ci, sw * si)
    d2 = (-sw * co - cw * so * ci, -sw * so + cw * co * ci, cw * si)
    return [d1[k] * xfac1 + d2[k] * xfac2 for k in range(3)] + [d1[k] * vfac1 + d2[k] * vfac2 for k in range(3)]


def find_max_energy_conservation_error(ens, reference_ensemble):
    """swarm/utils.cpp:105-121"""
    e0, e1 = reference_ensemble.calc_total_energy(), ens.calc_total_energy()
    return float(np.abs((e1 - e0) / e0).max())
