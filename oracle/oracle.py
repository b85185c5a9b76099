"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/swarm_oracle.h).  The product path (swarm_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

ACTIVE, INACTIVE, NEEDS_EXAM, DISABLED = 0, 1, 2, -1
MON_EJECTION, MON_LOG_INTERVAL, MON_EJECTION_OR_CLOSE, MON_MCE_STAT, MON_LOG_TRANSIT = 0, 1, 2, 3, 4
INTEGRATORS = {
    "hermite_cpu": 0, "hermite": 1, "hermite_adap": 2, "rkck_fixed": 3, "rkck_adaptive": 4,
    "mvs": 5, "euler": 6, "verlet": 7, "midpoint": 8, "hermite_prop": 9, "irk2_cpu": 10, "irk2": 11,
}
FLT_MAX = float(np.finfo(np.float32).max)
MAX_NBOD = 32  # ORC_MAX_NBOD, swarm_oracle.h


def build(force=False):
    """Compile the C restatement (gcc); building the checker is not using it."""
    src = os.path.join(_HERE, "swarm_oracle.c")
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(src),
                                              os.path.getmtime(os.path.join(_HERE, "swarm_oracle.h")))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


class _Ens(C.Structure):
    _fields_ = [("nsys", C.c_int32), ("nbod", C.c_int32), ("pos", C.c_void_p), ("vel", C.c_void_p),
                ("mass", C.c_void_p), ("time", C.c_void_p), ("state", C.c_void_p), ("id", C.c_void_p),
                ("nbattr", C.c_int32), ("nsattr", C.c_int32), ("battr", C.c_void_p), ("sattr", C.c_void_p)]


class _Mon(C.Structure):
    _fields_ = [("kind", C.c_int32), ("rmax", C.c_double), ("deactivate_on_ejection", C.c_int32),
                ("log_on_ejection", C.c_int32), ("verbose_on_ejection", C.c_int32),
                ("log_interval", C.c_double), ("close_approach", C.c_double),
                ("deactivate_on_close_encounter", C.c_int32), ("log_on_close_encounter", C.c_int32),
                ("verbose_on_close_encounter", C.c_int32), ("rceh", C.c_double),
                ("mce_time_step", C.c_double), ("tr_tol", C.c_double), ("tr_dt", C.c_double),
                ("tr_log_transit", C.c_int32), ("tr_log_occultation", C.c_int32)]


class _Log(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("cap", C.c_int64), ("len", C.c_int64), ("dropped", C.c_int64)]


class _Par(C.Structure):
    _fields_ = [("integrator", C.c_int32), ("max_iterations", C.c_int32),
                ("destination_time", C.c_double), ("time_step", C.c_double),
                ("time_step_factor", C.c_double), ("min_time_step", C.c_double),
                ("max_time_step", C.c_double), ("error_tolerance", C.c_double),
                ("max_timestep", C.c_double), ("timestep_scale", C.c_double), ("irk2_method", C.c_int32),
                ("kepler_compensated", C.c_int32)]


class _Cnt(C.Structure):
    _fields_ = [("accepted", C.c_void_p), ("attempted", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_launch.restype = C.c_int
        _lib.orc_integrate.restype = C.c_int
        _lib.orc_snapshot_record_size.restype = C.c_int64
        _lib.orc_generate.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    return _lib


class Ensemble:
    """Plain AoS host ensemble: pos/vel [nsys,nbod,3], mass [nsys,nbod], time/state/id [nsys].

    Attributes (body_attr [nsys,nbod,nbattr], sys_attr [nsys,nsattr]) ride along for file I/O only.
    """

    def __init__(self, nsys, nbod, nbattr=1, nsattr=1):
        self.nsys, self.nbod = int(nsys), int(nbod)
        self.pos = np.zeros((nsys, nbod, 3))
        self.vel = np.zeros((nsys, nbod, 3))
        self.mass = np.zeros((nsys, nbod))
        self.time = np.zeros(nsys)
        self.state = np.zeros(nsys, dtype=np.int32)
        self.id = np.arange(nsys, dtype=np.int32)
        self.body_attr = np.zeros((nsys, nbod, nbattr))
        self.sys_attr = np.zeros((nsys, nsattr))

    def clone(self):
        e = Ensemble(self.nsys, self.nbod, self.body_attr.shape[2], self.sys_attr.shape[1])
        for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
            setattr(e, k, getattr(self, k).copy())
        return e

    def _c(self):
        for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
            a = getattr(self, k)
            assert a.flags["C_CONTIGUOUS"] and a.dtype in (np.float64, np.int32), k
        return _Ens(self.nsys, self.nbod, self.pos.ctypes.data, self.vel.ctypes.data,
                    self.mass.ctypes.data, self.time.ctypes.data, self.state.ctypes.data,
                    self.id.ctypes.data, self.body_attr.shape[2], self.sys_attr.shape[1],
                    self.body_attr.ctypes.data if self.body_attr.size else None,
                    self.sys_attr.ctypes.data if self.sys_attr.size else None)


def generate(nsys, nbod, spacing_factor=1.4, planet_mass=1e-4, ejection_factor=1.0 / np.sqrt(2),
             star_radius=0.1, planet_radius=0.001):
    """swarm/utils.cpp:63-103"""
    e = Ensemble(nsys, nbod)
    c = e._c()
    lib().orc_generate(C.byref(c), spacing_factor, planet_mass, ejection_factor)
    e.body_attr[:, 0, 0] = star_radius
    e.body_attr[:, 1:, 0] = planet_radius
    return e


def monitor(kind=MON_EJECTION, rmax=FLT_MAX, deactivate_on_ejection=0, log_on_ejection=0,
            verbose_on_ejection=0, log_interval=0.0, close_approach=0.0,
            deactivate_on_close_encounter=0, log_on_close_encounter=0, verbose_on_close_encounter=0,
            rceh=1e-6, mce_time_step=0.0, log_transit_tol=2e-8, transit_dt=0.0, log_transits=1, log_occultations=0):
    return _Mon(kind, rmax, deactivate_on_ejection, log_on_ejection, verbose_on_ejection,
                log_interval, close_approach, deactivate_on_close_encounter, log_on_close_encounter,
                verbose_on_close_encounter, rceh, mce_time_step, log_transit_tol, transit_dt, log_transits,
                log_occultations)


def params(integrator, destination_time, max_iterations=10000000, time_step=0.0, time_step_factor=0.0,
           min_time_step=0.0, max_time_step=0.0, error_tolerance=0.0, max_timestep=0.0,
           timestep_scale=0.0, method=0, kepler_compensated=0):
    return _Par(INTEGRATORS[integrator], max_iterations, destination_time, time_step, time_step_factor,
                min_time_step, max_time_step, error_tolerance, max_timestep, timestep_scale, method,
                int(kepler_compensated))


class Log:
    def __init__(self, cap=50 << 20):
        self.mem = np.zeros(cap, dtype=np.uint8)
        self.c = _Log(self.mem.ctypes.data, cap, 0, 0)

    def bytes(self):
        return self.mem[: self.c.len].tobytes()

    def records(self):
        return parse_log(self.bytes())


def parse_log(raw):
    """Decode EVT_SNAPSHOT records -> list of dicts."""
    out, off = [], 0
    while off < len(raw):
        msgid, ln = np.frombuffer(raw, dtype=np.int32, count=2, offset=off)
        assert ln > 0 and ln % 16 == 0, (off, ln)
        rec = {"msgid": int(msgid), "len": int(ln)}
        if msgid == 1:
            rec["time"] = float(np.frombuffer(raw, dtype=np.float64, count=1, offset=off + 8)[0])
            sid, st, nb = np.frombuffer(raw, dtype=np.int32, count=3, offset=off + 16)
            rec.update(sys=int(sid), state=int(st), nbod=int(nb))
            bod = np.frombuffer(raw, dtype=np.float64, count=8 * nb, offset=off + 32).reshape(nb, 8)
            rec["bodies"] = bod[:, :7].copy()
            rec["body_id"] = np.frombuffer(raw, dtype=np.int32, count=16 * nb,
                                           offset=off + 32).reshape(nb, 16)[:, 14].copy()
        elif msgid in (15, 16):                # EVT_TRANSIT / EVT_OCCULTATION, log/log.hpp:215-217, 261-265
            rec["time"] = float(np.frombuffer(raw, dtype=np.float64, count=1, offset=off + 8)[0])
            sid, bod = np.frombuffer(raw, dtype=np.int32, count=2, offset=off + 16)
            b, vproj = np.frombuffer(raw, dtype=np.float64, count=2, offset=off + 24)
            rec.update(sys=int(sid), body=int(bod), b=float(b), vproj=float(vproj))
        elif msgid == -2:                      # MSG_PRINTF, gpulog_constants.h:35
            rec["text"] = run_printf(raw[off:off + int(ln)])
        out.append(rec)
        off += int(ln)
    return out


def run_printf(rec):
    """lprintf.h:74-171 run_printf: format string at offset 8, arguments at natural alignment."""
    import re
    end = rec.index(b"\0", 8)
    fmt = rec[8:end].decode()
    at, args = end + 1, []
    for m in re.finditer(r"%[^%cdiouxXeEfgGaAnps]*([%cdiouxXeEfgGaAnps])", fmt):
        k = m.group(1)
        if k in "cdiouxXp":
            at = (at + 3) // 4 * 4
            args.append(int(np.frombuffer(rec, dtype=np.int32, count=1, offset=at)[0]))
            at += 4
        elif k in "eEfgGaA":
            at = (at + 7) // 8 * 8
            args.append(float(np.frombuffer(rec, dtype=np.float64, count=1, offset=at)[0]))
            at += 8
    return fmt.replace("%lg", "%g") % tuple(args)


def integrate(ens, par, mon=None, log=None, threads=1, max_attempts=1000000, counters=False,
              single_launch=False):
    """swarm::integrator::integrate (integrator.cpp:91-100) over the oracle kernels, in place."""
    mon = mon or monitor()
    c = ens._c()
    acc = np.zeros(ens.nsys, dtype=np.int64)
    att = np.zeros(ens.nsys, dtype=np.int64)
    cnt = _Cnt(acc.ctypes.data, att.ctypes.data)
    lg = C.byref(log.c) if log is not None else None
    if single_launch:
        rc = lib().orc_launch(C.byref(c), C.byref(par), C.byref(mon), lg, C.byref(cnt), int(threads))
    else:
        rc = lib().orc_integrate(C.byref(c), C.byref(par), C.byref(mon), lg, C.byref(cnt),
                                 int(threads), int(max_attempts))
    if rc < 0:
        raise RuntimeError("oracle launch failed: %d" % rc)
    if counters:
        return acc, att
    return rc


def total_energy(ens):
    out = np.zeros(ens.nsys)
    c = ens._c()
    lib().orc_total_energy(C.byref(c), out.ctypes.data_as(C.c_void_p))
    return out


def energy_error(ens, ref):
    """energy_conservation_error_range, utils.cpp:108-121 -> per-system |dE/E0|"""
    e0, e1 = total_energy(ref), total_energy(ens)
    return np.abs((e1 - e0) / e0)


def compare(a, b):
    """compare_ensembles, utils.cpp:170-214 -> (pos_diff, vel_diff, time_diff)"""
    pd, vd, td = C.c_double(), C.c_double(), C.c_double()
    ca, cb = a._c(), b._c()
    ok = lib().orc_compare(C.byref(ca), C.byref(cb), C.byref(pd), C.byref(vd), C.byref(td))
    assert ok
    return pd.value, vd.value, td.value


def pair(nbod, ij):
    return lib().orc_pair_first(nbod, ij), lib().orc_pair_second(nbod, ij)


def max_threads():
    return lib().orc_max_threads()


# ---- text snapshot format, swarm/snapshot.cpp:95-216 ("SwarmDataFile 2") -------------------
def load_text(path):
    tok = open(path).read().split()
    assert tok[0] == "SwarmDataFile" and int(tok[1]) == 2, "Invalid file, header doesn't match"
    nbod, nsys, nsattr, nbattr = (int(t) for t in tok[2:6])
    e = Ensemble(nsys, nbod, nbattr, nsattr)
    k = 6
    for i in range(nsys):
        e.id[i], e.time[i], e.state[i] = int(tok[k]), float(tok[k + 1]), int(tok[k + 2])
        k += 3
        for a in range(nsattr):
            e.sys_attr[i, a] = float(tok[k]); k += 1
        for b in range(nbod):
            v = [float(t) for t in tok[k:k + 7 + nbattr]]
            k += 7 + nbattr
            e.mass[i, b] = v[0]
            e.pos[i, b] = v[1:4]
            e.vel[i, b] = v[4:7]
            e.body_attr[i, b] = v[7:]
    assert k == len(tok)
    return e


def save_text(e, path):
    with open(path, "w") as f:
        f.write("SwarmDataFile 2\n%i %i %i %i\n\n\n" % (e.nbod, e.nsys, e.sys_attr.shape[1],
                                                      e.body_attr.shape[2]))
        for i in range(e.nsys):
            f.write("%i %.15e %i\n" % (e.id[i], e.time[i], e.state[i]))
            f.write("".join("%.15e " % a for a in e.sys_attr[i]) + "\n")
            for b in range(e.nbod):
                f.write("\t%.15e\n\t%.15e %.15e %.15e\n\t%.15e %.15e %.15e\n\t" % (
                    e.mass[i, b], *e.pos[i, b], *e.vel[i, b]))
                f.write("".join("%.15e " % a for a in e.body_attr[i, b]) + "\n\n")
            f.write("\n")
