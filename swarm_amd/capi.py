"""ctypes view of include/swarm_hip.h (libswarm_hip.so).  No fallback: if the HIP library is
missing, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWARM_HIP_LIB: explicit path of the C-ABI library (kernel-variant experiments, packaging)
LIB_PATH = os.environ.get("SWARM_HIP_LIB") or os.path.join(_HERE, "lib", "libswarm_hip.so")

CHUNK = 64
NOALLOC_MAX_NSYS = 1 << 26
MIN_NBOD, MAX_NBOD, MAX_NBOD_LANE_PER_SYSTEM = 3, 32, 10
ACTIVE, INACTIVE, NEEDS_EXAM, DISABLED = 0, 1, 2, -1
HERMITE, HERMITE_ADAP, RKCK_FIXED, RKCK_ADAPTIVE, MVS, EULER, VERLET, MIDPOINT, HERMITE_PROP, IRK2, IRK2_CPU = range(1, 12)
MON_EJECTION, MON_LOG_INTERVAL, MON_EJECTION_OR_CLOSE, MON_MCE_STAT, MON_LOG_TRANSIT = 0, 1, 2, 3, 4


class SwarmHipError(RuntimeError):
    pass


class EnsembleStruct(C.Structure):
    _fields_ = [("nsys", C.c_int32), ("nbod", C.c_int32), ("nbattr", C.c_int32), ("nsattr", C.c_int32),
                ("body", C.c_void_p), ("sysd", C.c_void_p), ("sysi", C.c_void_p)]


class LogStruct(C.Structure):
    _fields_ = [("d_buf", C.c_void_p), ("d_cursor", C.c_void_p), ("capacity", C.c_int32)]


class LaunchStruct(C.Structure):
    _fields_ = [("integrator", C.c_int32), ("monitor", C.c_int32), ("max_iterations", C.c_int32),
                ("block_size", C.c_int32), ("destination_time", C.c_double), ("time_step", C.c_double),
                ("time_step_factor", C.c_double), ("min_time_step", C.c_double),
                ("max_time_step", C.c_double), ("error_tolerance", C.c_double),
                ("max_timestep", C.c_double), ("timestep_scale", C.c_double), ("rmax", C.c_double),
                ("deactivate_on_ejection", C.c_int32), ("log_on_ejection", C.c_int32),
                ("verbose_on_ejection", C.c_int32), ("deactivate_on_close_encounter", C.c_int32),
                ("log_on_close_encounter", C.c_int32), ("verbose_on_close_encounter", C.c_int32),
                ("close_approach", C.c_double), ("log_interval", C.c_double), ("rceh", C.c_double), ("irk2_method", C.c_int32), ("log_transits", C.c_int32), ("log_occultations", C.c_int32),
                ("lane_mode", C.c_int32), ("log_transit_tol", C.c_double),
                ("kepler_compensated", C.c_int32), ("active_hint", C.c_int32),
                ("log", LogStruct),
                ("d_steps_accepted", C.c_void_p), ("d_steps_attempted", C.c_void_p),
                ("stream", C.c_void_p)]


# every symbol include/swarm_hip.h declares
SYMBOLS = [
    "swarm_hip_init", "swarm_hip_last_error", "swarm_hip_version", "swarm_hip_nchunk",
    "swarm_hip_body_bytes", "swarm_hip_sysd_bytes", "swarm_hip_sysi_bytes",
    "swarm_hip_snapshot_record_bytes", "swarm_hip_launch_integrator", "swarm_hip_count_active",
    "swarm_hip_count_active_noalloc",
    "swarm_hip_reactivate", "swarm_hip_total_energy", "swarm_hip_energy_stats",
    "swarm_hip_import_ref_layout", "swarm_hip_export_ref_layout", "swarm_hip_upload_aos",
    "swarm_hip_download_aos", "swarm_hip_malloc", "swarm_hip_free", "swarm_hip_memset",
    "swarm_hip_memcpy_d2h", "swarm_hip_memcpy_h2d", "swarm_hip_synchronize", "swarm_hip_host_alloc",
    "swarm_hip_host_free", "swarm_hip_event_create",
    "swarm_hip_event_record", "swarm_hip_event_elapsed_ms", "swarm_hip_event_destroy",
    "swarm_hip_event_synchronize", "swarm_hip_stream_create", "swarm_hip_stream_destroy",
    "swarm_hip_stream_wait_event", "swarm_hip_memcpy_d2h_async",
]

if not os.path.exists(LIB_PATH):
    raise ImportError("libswarm_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C swarm_amd/csrc` (expected at %s)" % LIB_PATH)
lib = C.CDLL(LIB_PATH)

lib.swarm_hip_last_error.restype = C.c_char_p
lib.swarm_hip_version.restype = C.c_char_p
lib.swarm_hip_nchunk.restype = C.c_int64
lib.swarm_hip_nchunk.argtypes = [C.c_int32]
lib.swarm_hip_body_bytes.restype = C.c_size_t
lib.swarm_hip_body_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
lib.swarm_hip_sysd_bytes.restype = C.c_size_t
lib.swarm_hip_sysd_bytes.argtypes = [C.c_int32, C.c_int32]
lib.swarm_hip_sysi_bytes.restype = C.c_size_t
lib.swarm_hip_sysi_bytes.argtypes = [C.c_int32]
lib.swarm_hip_snapshot_record_bytes.restype = C.c_int64
lib.swarm_hip_snapshot_record_bytes.argtypes = [C.c_int32]
lib.swarm_hip_init.argtypes = [C.c_int]
lib.swarm_hip_launch_integrator.argtypes = [C.POINTER(EnsembleStruct), C.POINTER(LaunchStruct)]
lib.swarm_hip_count_active.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p, C.POINTER(C.c_int32)]
lib.swarm_hip_count_active_noalloc.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p, C.c_void_p]
lib.swarm_hip_reactivate.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p]
lib.swarm_hip_total_energy.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p, C.c_void_p]
lib.swarm_hip_energy_stats.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.swarm_hip_import_ref_layout.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
lib.swarm_hip_export_ref_layout.argtypes = [C.POINTER(EnsembleStruct), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
lib.swarm_hip_upload_aos.argtypes = [C.POINTER(EnsembleStruct)] + [C.c_void_p] * 8
lib.swarm_hip_download_aos.argtypes = [C.POINTER(EnsembleStruct)] + [C.c_void_p] * 8
lib.swarm_hip_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
lib.swarm_hip_free.argtypes = [C.c_void_p]
lib.swarm_hip_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
lib.swarm_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.swarm_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.swarm_hip_synchronize.argtypes = [C.c_void_p]
lib.swarm_hip_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
lib.swarm_hip_host_free.argtypes = [C.c_void_p]
lib.swarm_hip_event_create.argtypes = [C.POINTER(C.c_void_p)]
lib.swarm_hip_event_record.argtypes = [C.c_void_p, C.c_void_p]
lib.swarm_hip_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
lib.swarm_hip_event_destroy.argtypes = [C.c_void_p]
lib.swarm_hip_event_synchronize.argtypes = [C.c_void_p]
lib.swarm_hip_stream_create.argtypes = [C.POINTER(C.c_void_p)]
lib.swarm_hip_stream_destroy.argtypes = [C.c_void_p]
lib.swarm_hip_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
lib.swarm_hip_memcpy_d2h_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]


def check(rc):
    """Convert a C-ABI return code into an exception (the role of swarm::runtime_error /
    cudaException, swarm/runtime_error.hpp:38-74)."""
    if rc != 0:
        raise SwarmHipError("swarm_hip error %d: %s" % (rc, lib.swarm_hip_last_error().decode()))
    return rc
