"""Host side of the integrator path, mirroring the reference's operator surface:

  swarm::plugin registry          swarm/plugin.{hpp,cpp}          -> PLUGINS / plugin_not_found
  swarm::integrator::create(cfg)  swarm/integrator.cpp:129-145    -> Integrator.create(cfg)
  gpu::integrator                 swarm/integrator.{hpp,cpp}      -> GpuIntegrator
  log::manager                    swarm/log/logmanager.{hpp,cpp}  -> LogManager

Every GPU plugin's launch_integrator() is ONE call of swarm_hip_launch_integrator()."""
import ctypes as C

import numpy as np

import sys

from . import capi, gpulog
from .config import config, key_not_found  # noqa: F401
from .ensemble import FLT_MAX, DeviceBuffer, DeviceEnsemble, HostEnsemble


class plugin_not_found(RuntimeError):
    """swarm/plugin.hpp:187-196"""


class runtime_error(RuntimeError):
    """swarm::runtime_error via ERROR(), swarm/runtime_error.hpp:38-51"""


# plugin id -> (C-ABI integrator, C-ABI monitor); names as registered in src/plugins/*.cu.
# `hermite` is bound to mce_stat in this fork (plugins/hermite.cu:40).  That monitor reads and
# writes body attributes 1 and 2, which do not exist in the reference's default build
# (NUM_PLANET_ATTRIBUTES=1: the accesses run past the Body struct).  Here `hermite` binds mce_stat
# when the ensemble carries >= 3 body attributes -- the only case where the reference's behaviour
# is defined -- and otherwise the stop_on_ejection monitor of hermite_cpu, the parity oracle.
PLUGINS = {
    "integrator_hermite": (capi.HERMITE, capi.MON_EJECTION),
    # the reference's CPU plugin names (plugins/hermite_cpu.cpp:36-56), served by the same HIP kernel:
    # its arithmetic follows hermite_cpu and reproduces hermite_cpu's golden outputs
    "integrator_hermite_cpu": (capi.HERMITE, capi.MON_EJECTION),
    "integrator_hermite_cpu_ejection_or_close_encounter": (capi.HERMITE, capi.MON_EJECTION_OR_CLOSE),
    "integrator_hermite_cpu_log": (capi.HERMITE, capi.MON_LOG_INTERVAL),
    "integrator_hermite_adap": (capi.HERMITE_ADAP, capi.MON_EJECTION),
    "integrator_hermite_adap_log": (capi.HERMITE_ADAP, capi.MON_LOG_INTERVAL),
    "integrator_hermite_adap_close_encounter": (capi.HERMITE_ADAP, capi.MON_EJECTION_OR_CLOSE),
    "integrator_hermite_adap_transit": (capi.HERMITE_ADAP, capi.MON_LOG_TRANSIT),      # plugins/hermite_adap.cu:61
    "integrator_rkck_fixed": (capi.RKCK_FIXED, capi.MON_EJECTION),
    "integrator_rkck_fixed_close_encounter": (capi.RKCK_FIXED, capi.MON_EJECTION_OR_CLOSE),
    "integrator_rkck_adaptive": (capi.RKCK_ADAPTIVE, capi.MON_EJECTION),
    "integrator_rkck_adaptive_close_encounter": (capi.RKCK_ADAPTIVE, capi.MON_EJECTION_OR_CLOSE),
    "integrator_mvs": (capi.MVS, capi.MON_EJECTION),
    "integrator_mvs_close_encounter": (capi.MVS, capi.MON_EJECTION_OR_CLOSE),
    "integrator_euler": (capi.EULER, capi.MON_EJECTION),
    "integrator_verlet": (capi.VERLET, capi.MON_EJECTION),
    "integrator_midpoint": (capi.MIDPOINT, capi.MON_EJECTION),
    "integrator_hermite_prop": (capi.HERMITE_PROP, capi.MON_EJECTION),
    # Gauss implicit Runge-Kutta: plugins/irk2.cu:39 (bound to mce_stat like `hermite`), plugins/irk2_cpu.cpp:36
    "integrator_irk2": (capi.IRK2, capi.MON_EJECTION),
    "integrator_irk2_cpu": (capi.IRK2_CPU, capi.MON_EJECTION),
}


class BinaryWriter:
    """log::binary_writer (swarm/log/binary_writer.cpp:39-91): the raw record stream behind a 96-byte
    swarm_header in `<log_output>.raw`; close() sorts it by (time, system, position) into `<log_output>`
    (swarm/log/io.cpp:226-264) and removes the raw file.  Same bytes as the C++ host mirror writes."""

    UNSORTED = b"unsorted_output // Unsorted output file"
    SORTED = b"T_sorted_output // Output file sorted by time"

    @staticmethod
    def header(kind, datalen=0xFFFFFFFFFFFFFFFF):
        import struct
        return struct.pack("<6s2s76sIQ", b"SWARM", b"0", kind, 0, datalen)   # fileformat.hpp:28-68

    def __init__(self, cfg):
        self.binfn = cfg.optional("log_output", "")
        if self.binfn == "":
            raise runtime_error("Expected filename for writer.")
        self.rawfn = self.binfn + ".raw"
        self._f = open(self.rawfn, "wb")
        self._f.write(self.header(self.UNSORTED))

    def process(self, data):
        if data:
            self._f.write(data)

    def close(self):
        if self._f is None:
            return
        self._f.close()
        self._f = None
        import os
        import struct
        raw = open(self.rawfn, "rb").read()[96:]
        recs = []
        for pos, (msgid, rec) in enumerate(gpulog.records(raw)):
            # gpulog's own records (msgid < 0: lprintf lines) have no (T, sys) heading and sort first
            # with the key (-1, -1), swarm/log/io.cpp:46-58
            t, s = struct.unpack_from("<di", rec, 8) if (msgid >= 0 and len(rec) >= 20) else (-1.0, -1)
            recs.append((t, s, pos, rec))
        recs.sort(key=lambda r: r[:3])
        with open(self.binfn, "wb") as out:
            out.write(self.header(self.SORTED, len(raw)))
            for r in recs:
                out.write(r[3])
        os.unlink(self.rawfn)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LogManager:
    """log::manager (swarm/log/logmanager.cpp:50-87): owns the device event buffer; flush() copies
    the used bytes to the host and clears the device cursor.  The default log has 0-byte buffers
    and a null writer (logmanager.cpp:37-47).

    buffers=2 (no counterpart in the reference, whose flush is one synchronous copy after every launch): two device
    logs and two page-locked staging areas.  flush_async() starts the copy of the log the last launch filled on a
    second stream and hands the next launch the other log, so that the copy runs while that launch computes; the
    writer sees the same records (their order in the stream is free in the reference too: one atomic cursor)."""

    def __init__(self, device_bytes=0, writer=None, buffers=1):
        self.capacity = int(device_bytes)
        self.writer = writer  # callable(bytes) or None (= null writer)
        self.nbuf = int(buffers) if self.capacity else 1
        assert self.nbuf in (1, 2)
        self.bufs = [DeviceBuffer(self.capacity) for _ in range(self.nbuf)] if self.capacity else []
        self.cursors = [DeviceBuffer(16) for _ in range(self.nbuf)] if self.capacity else []  # int32[4]: used, dropped, lprintf records, -
        for c in self.cursors:
            c.zero()
        self.cur = 0                # the log the next launch appends to
        self.host = bytearray()
        self.replay_printf = True   # lprintf lines go to stderr at every flush (logmanager.cpp:79-83)
        self.printf_count = 0
        self.dropped = 0
        self.bytes_flushed = 0
        # pinned staging buffers: the D2H copy of a flush runs at PCIe speed
        self._pinneds, self._stages = [], []
        for _ in range(self.nbuf if self.capacity else 0):
            p = C.c_void_p()
            capi.check(capi.lib.swarm_hip_host_alloc(C.byref(p), self.capacity))
            self._pinneds.append(p)
            self._stages.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.capacity,)))
        # asynchronous flush: copy stream, one event behind the launch, one behind each copy, a pinned cursor image
        self._copy_stream = C.c_void_p()
        self._pending = [None] * self.nbuf     # (used, dropped, nprintf) of a copy in flight, per log
        self._dirty = [False] * self.nbuf      # the log's device cursor still holds the count of a copy that was started
        self._ev_launch = C.c_void_p()
        self._ev_copy = [C.c_void_p() for _ in range(self.nbuf)]
        self._pinned_cur = C.c_void_p()
        if self.nbuf == 2:
            capi.check(capi.lib.swarm_hip_stream_create(C.byref(self._copy_stream)))
            capi.check(capi.lib.swarm_hip_event_create(C.byref(self._ev_launch)))
            for e in self._ev_copy:
                capi.check(capi.lib.swarm_hip_event_create(C.byref(e)))
            capi.check(capi.lib.swarm_hip_host_alloc(C.byref(self._pinned_cur), 16))
            self._cur_img = np.ctypeslib.as_array(C.cast(self._pinned_cur, C.POINTER(C.c_int32)), shape=(4,))

    # the single-log names, for callers that look at the device log directly
    @property
    def buf(self):
        return self.bufs[self.cur] if self.bufs else None

    @property
    def cursor(self):
        return self.cursors[self.cur] if self.cursors else None

    @property
    def _pinned(self):
        return self._pinneds[self.cur] if self._pinneds else C.c_void_p()

    @property
    def _stage(self):
        return self._stages[self.cur] if self._stages else None

    def __del__(self):
        try:
            L = capi.lib
            if self._copy_stream:
                L.swarm_hip_synchronize(self._copy_stream)
                L.swarm_hip_stream_destroy(self._copy_stream)
                self._copy_stream = C.c_void_p()
                for e in [self._ev_launch] + self._ev_copy:
                    L.swarm_hip_event_destroy(e)
                self._cur_img = None
                L.swarm_hip_host_free(self._pinned_cur)
            self._stages = []
            for p in self._pinneds:
                L.swarm_hip_host_free(p)
            self._pinneds = []
        except Exception:
            pass

    @staticmethod
    def from_config(cfg, device_bytes=50 << 20):
        """manager::init(cfg, host_sz, dev_sz), logmanager.cpp:50-58; `log_writer` selects the sink"""
        w = cfg.optional("log_writer", "null")
        if w == "null":
            return LogManager(device_bytes, None)
        if w == "host_array":
            return LogManager(device_bytes, "host")
        if w == "binary":
            return LogManager(device_bytes, BinaryWriter(cfg))
        raise plugin_not_found("writer_" + w)

    def close(self):
        """flush what is left and finish the writer (the binary writer sorts its file here)"""
        self.flush()
        if hasattr(self.writer, "close"):
            self.writer.close()

    def struct(self):
        if not self.capacity:
            return capi.LogStruct(None, None, 0)
        return capi.LogStruct(self.bufs[self.cur].ptr, self.cursors[self.cur].ptr, self.capacity)

    def _process(self, stage, used, dropped, nprintf):
        """what manager::flush does with the host copy (logmanager.cpp:79-86): replay the printf records, hand the
        bytes to the writer"""
        self.dropped += dropped
        if used:
            self.bytes_flushed += used
            if self.replay_printf and nprintf:   # logmanager.cpp:83: replay_printf(std::cerr, hlog)
                self.printf_count += gpulog.replay_printf(sys.stderr, stage[:used].tobytes())
            if self.writer == "host":
                self.host += memoryview(stage[:used])
            elif self.writer == "count":
                pass  # bytes_flushed only: measures the flush protocol without a sink
            elif self.writer is not None:
                self.writer.process(stage[:used].tobytes())

    def _finish(self, i):
        """wait for the copy of log i that flush_async started, and process its bytes"""
        if self._pending[i] is not None:
            capi.check(capi.lib.swarm_hip_event_synchronize(self._ev_copy[i]))
            pend, self._pending[i] = self._pending[i], None
            self._process(self._stages[i], *pend)

    def flush(self):
        """Synchronous flush (the reference's protocol).  The launch stream must have drained."""
        if not self.capacity:
            return
        for i in range(self.nbuf):
            self._finish(i)
            if self._dirty[i]:
                self.cursors[i].zero()
                self._dirty[i] = False
        cur = self.cursors[self.cur].to_host(np.int32)
        used, dropped, nprintf = int(cur[0]), int(cur[1]), int(cur[2])
        if used:
            capi.check(capi.lib.swarm_hip_memcpy_d2h(self._pinneds[self.cur], self.bufs[self.cur].ptr, used))
        self._process(self._stages[self.cur], used, dropped, nprintf)
        if used or dropped:
            self.cursors[self.cur].zero()

    def flush_async(self, launch_stream=None):
        """Start flushing the log that the launches enqueued so far on `launch_stream` have filled, and switch to the
        other log: the next launch may be enqueued as soon as this returns and runs while the copy is in flight.
        Returns after (1) those launches have finished -- the copy's length is the cursor they leave -- and (2) the
        previous copy out of the log that is current from now on has been handed to the writer.  Nothing is lost when
        a copy overlaps a running kernel: the kernel appends to the other log."""
        if self.nbuf != 2:
            if launch_stream:
                capi.check(capi.lib.swarm_hip_synchronize(launch_stream))
            return self.flush()
        L, i, cs = capi.lib, self.cur, self._copy_stream
        # the cursor travels on the LAUNCH stream, right behind the kernels: the host learns it the moment they finish,
        # while the previous log's copy is still in flight on the copy stream, and queues this log's copy behind that
        # one without a gap
        capi.check(L.swarm_hip_memcpy_d2h_async(self._pinned_cur, self.cursors[i].ptr, 16, launch_stream))
        capi.check(L.swarm_hip_event_record(self._ev_launch, launch_stream))
        capi.check(L.swarm_hip_event_synchronize(self._ev_launch))
        used, dropped, nprintf = int(self._cur_img[0]), int(self._cur_img[1]), int(self._cur_img[2])
        # (the kernels are known to have finished: the copy needs no stream-side wait, and nothing but the copy and
        # its event goes onto the copy stream, so that successive copies follow each other without a gap)
        if used:
            capi.check(L.swarm_hip_memcpy_d2h_async(self._pinneds[i], self.bufs[i].ptr, used, cs))
        capi.check(L.swarm_hip_event_record(self._ev_copy[i], cs))
        self._pending[i] = (used, dropped, nprintf)
        self._dirty[i] = bool(used or dropped)
        self.cur = 1 - i
        self._finish(self.cur)
        if self._dirty[self.cur]:   # cleared on the launch stream, ahead of the launch that appends to it next
            capi.check(L.swarm_hip_memset(self.cursors[self.cur].ptr, 0, 16, launch_stream))
            self._dirty[self.cur] = False


class Integrator:
    """swarm::integrator surface (swarm/integrator.hpp:49-149)."""

    _default_max_iterations = 10000000
    _default_max_attempts = 1000000

    @staticmethod
    def create(cfg):
        if "integrator" not in cfg:
            raise runtime_error("Integrator type has to be chosen with 'integrator=xxx' keyword")
        name = cfg["integrator"]
        if "integrator_" + name not in PLUGINS:
            raise runtime_error("Integrator " + name + " not found.")
        return GpuIntegrator(cfg)


class GpuIntegrator(Integrator):
    """gpu::integrator (swarm/integrator.hpp:178-275, integrator.cpp:102-117)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.name = cfg["integrator"]
        self.int_id, self.mon_id = PLUGINS["integrator_" + self.name]
        self._max_iterations = cfg.optional("max_iterations", self._default_max_iterations)
        self._max_attempts = cfg.optional("max_attempts", self._default_max_attempts)
        self._destination_time = 0.0
        self._hens = None
        self._dens = None
        self._logman = LogManager()
        self.stream = None
        a = capi.LaunchStruct()
        a.integrator, a.monitor = self.int_id, self.mon_id
        a.block_size = 0
        spb = cfg.optional("system_per_block", 0)
        if spb:
            # one wavefront or four per workgroup: two or three measured up to 2x slower at one wavefront per SIMD (swarm_hip.h)
            a.block_size = 64 if spb <= 64 else 256
        # constructor-time key checks follow each reference ctor
        if self.int_id in (capi.HERMITE, capi.MVS, capi.EULER, capi.MIDPOINT, capi.HERMITE_PROP, capi.IRK2, capi.IRK2_CPU):
            a.time_step = cfg.require("time_step", 0.0)      # hermite_integrator.hpp:44-46 etc.
        if self.int_id == capi.HERMITE_ADAP:
            a.time_step_factor = cfg.require("time_step_factor", 0.0)  # hermite_adap.hpp:45-46
            a.min_time_step = cfg.require("min_time_step", 0.0)
        if self.int_id in (capi.RKCK_FIXED, capi.RKCK_ADAPTIVE):     # rkck_integrator.hpp:63-71
            for k, msg in (("min_time_step", "min timestep ('min time step' keyword in the config file)"),
                           ("max_time_step", "max timestep ('max time step' keyword in the config file)"),
                           ("error_tolerance", "error tolerance ('error tolerance' keyword in the config file)")):
                if k not in cfg:
                    raise runtime_error("Integrator rkck requires a " + msg + ".")
            a.min_time_step = _atof(cfg["min_time_step"])
            a.max_time_step = _atof(cfg["max_time_step"])
            a.error_tolerance = _atof(cfg["error_tolerance"])
        if self.int_id == capi.VERLET:                                # verlet.hpp:35-39
            a.max_timestep = cfg.require("max_timestep", 0.0)
            a.timestep_scale = cfg.require("timestep_scale", 0.0)
        # monitor params
        if self.mon_id in (capi.MON_EJECTION, capi.MON_EJECTION_OR_CLOSE):  # stop_on_ejection.hpp:47-53
            a.rmax = cfg.optional("rmax", FLT_MAX)
            a.deactivate_on_ejection = int(cfg.optional("deactivate_on_ejection", False))
            a.log_on_ejection = int(cfg.optional("log_on_ejection", False))
            a.verbose_on_ejection = int(cfg.optional("verbose_on_ejection", False))
        if self.mon_id == capi.MON_EJECTION_OR_CLOSE:            # stop_on_close_encounter.hpp:43-49
            a.close_approach = cfg.optional("close_approach", 0.0)
            a.deactivate_on_close_encounter = int(cfg.optional("deactivate_on_close_encounter", False))
            a.log_on_close_encounter = int(cfg.optional("log_on_close_encounter", False))
            a.verbose_on_close_encounter = int(cfg.optional("verbose_on_close_encounter", False))
        if self.mon_id == capi.MON_LOG_INTERVAL:                      # log_time_interval.hpp:39-43
            a.log_interval = cfg.require("log_interval", 0.0)
        if self.mon_id == capi.MON_LOG_TRANSIT:                       # log_transit_params, log_transit.hpp:48-58
            cfg.require("nbod", 0)
            a.log_transit_tol = cfg.optional("log_transit_tol", 2.e-8)
            a.time_step = cfg.require("time_step", 0.0)               # "use hermite's timestep"
            a.log_transits = int(cfg.optional("log_transits", True))
            a.log_occultations = int(cfg.optional("log_occultations", False))
        if self.name in ("hermite", "irk2"):                          # mce_stat_params, mercury_mce_stat.hpp:42-53
            a.rceh = cfg.optional("rceh", 1e-6)
        if self.int_id == capi.IRK2:                                  # irk2_integrator.hpp:53
            a.irk2_method = cfg.optional("method", 6)
        if self.int_id == capi.MVS:                                   # not a reference key, see include/swarm_hip.h
            a.kepler_compensated = int(cfg.optional("kepler_compensated", 0))
        a.lane_mode = cfg.optional("lane_mode", 0)                     # not a reference key: 0 auto, 1 per system, 2 per body
        self.use_active_hint = bool(cfg.optional("active_hint", 1))   # not a reference key: 0 = never pass the count on
        # not a reference key: with a two-log manager (LogManager(..., buffers=2)) every attempt of the launch loop runs
        # as this many launches over consecutive runs of wave chunks, the flush of one overlapping the launch of the next
        self.log_batches = max(1, int(cfg.optional("log_batches", 1)))
        self.args = a
        self.steps_accepted = None
        self.steps_attempted = None

    # ---- swarm::integrator API ----
    def set_ensemble(self, host, device=None):
        """Shares the caller's host ensemble: results appear in it after integrate()
        (integrator.hpp:233-236)."""
        self._hens = host
        self._dens = device or DeviceEnsemble(host.nsys, host.nbod, host.nbattr, host.nsattr)

    def get_ensemble(self):
        return self._hens

    def set_destination_time(self, t):
        self._destination_time = float(t)

    def get_destination_time(self):
        return self._destination_time

    def set_max_iterations(self, n):
        self._max_iterations = int(n)

    def set_max_attempts(self, n):
        self._max_attempts = int(n)

    def set_log_manager(self, lm):
        self._logman = lm

    def enable_step_counters(self):
        n = capi.lib.swarm_hip_nchunk(self._dens.nsys) * capi.CHUNK
        self.steps_accepted = DeviceBuffer(max(8 * n, 8))
        self.steps_attempted = DeviceBuffer(max(8 * n, 8))
        self.steps_accepted.zero()
        self.steps_attempted.zero()

    def step_counts(self):
        n = self._dens.nsys
        return (self.steps_accepted.to_host(np.uint64)[:n].copy(),
                self.steps_attempted.to_host(np.uint64)[:n].copy())

    def upload_ensemble(self):
        self._dens.upload(self._hens)

    def download_ensemble(self):
        self._dens.download(self._hens)

    def flush_log(self):
        self._logman.flush()

    def launch_integrator(self, view=None):
        """one launch over the whole device ensemble, or over `view` = self._dens.chunk_view(c0, c1)"""
        a = self.args
        a.destination_time = self._destination_time
        a.max_iterations = self._max_iterations
        a.log = self._logman.struct()
        first = view.first_system if view is not None else 0       # the step counters are indexed by system
        a.d_steps_accepted = self.steps_accepted.ptr + 8 * first if self.steps_accepted else None
        a.d_steps_attempted = self.steps_attempted.ptr + 8 * first if self.steps_attempted else None
        a.stream = self.stream
        # the count the attempt loop made after the previous launch (integrator.cpp:113), if nothing has touched the
        # states since: lets the library pick the kernel layout by ACTIVE systems and pack them into dense wavefronts
        a.active_hint = (self._dens.last_active_count or 0) if (self.use_active_hint and view is None) else 0
        if self.name in ("hermite", "irk2"):
            a.monitor = capi.MON_MCE_STAT if self._dens.nbattr >= 3 else capi.MON_EJECTION
            if a.monitor == capi.MON_MCE_STAT and "nbod" in self.cfg and int(self.cfg["nbod"]) != self._dens.nbod:
                raise runtime_error("mce_stat: nbod=%s in the configuration but the ensemble has %d bodies"
                                    % (self.cfg["nbod"], self._dens.nbod))
        capi.check(capi.lib.swarm_hip_launch_integrator((view or self._dens).ref, C.byref(a)))
        self._dens.last_active_count = None     # the launch changes states

    def core_integrate(self):
        """The device-resident part of integrate(): attempt loop without host<->device copies."""
        attempts = 0
        overlap = self._logman.nbuf == 2
        views = [None]
        if overlap and self.log_batches > 1:
            # the systems of an attempt in `log_batches` launches over consecutive runs of wave chunks: every system
            # is still launched once per attempt (same records, same counts), and the flush of one batch's log
            # overlaps the next batch's launch.  One launch per attempt otherwise.
            nchunk = int(capi.lib.swarm_hip_nchunk(self._dens.nsys))
            nb = min(self.log_batches, nchunk)
            cuts = [nchunk * k // nb for k in range(nb + 1)]
            views = [self._dens.chunk_view(cuts[k], cuts[k + 1]) for k in range(nb)]
        for _ in range(self._max_attempts):
            for v in views:
                self.launch_integrator(v)
                if overlap:     # the next launch (batch, or attempt) goes to the other log while this one is copied
                    self._logman.flush_async(self.stream)
                else:
                    if self.stream:     # the flush reads and clears the cursor on the null stream: let the launch stream drain first
                        capi.check(capi.lib.swarm_hip_synchronize(self.stream))
                    self._logman.flush()
            attempts += 1
            if self._dens.count_active(self.stream) == 0:
                break
        if overlap:
            self._logman.flush()     # the copies still in flight, and whatever the current log holds (nothing)
        return attempts

    def integrate(self):
        """gpu::integrator::integrate, integrator.cpp:102-117"""
        h = self._hens
        h.state[h.state == capi.INACTIVE] = capi.ACTIVE  # activate_inactive_systems, L83-89
        self.upload_ensemble()
        n = self.core_integrate()
        self.download_ensemble()
        return n


def _atof(s):
    from .config import _FLT
    m = _FLT.match(s)
    return float(m.group(1)) if m else 0.0
