/*
 * swarm_hip.h -- C-ABI of libswarm_hip.so: the MI355X (gfx950) ensemble-integrator hot path.
 *
 * This is the drop-in boundary for the reference's `gpu::integrator` path
 * (/root/reference/src, cited per entry point below).  Plain pointers and sizes only; no C++
 * or torch types cross it.  Every function returns 0 on success and a negative SWARM_HIP_E*
 * code on failure (never throws); swarm_hip_last_error() gives the message of the calling
 * thread's last failure.  All pointers named d_* / inside swarm_hip_ensemble_t are DEVICE
 * pointers on the current HIP device; `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Calls are asynchronous on `stream` unless stated otherwise.
 *
 * Device ensemble layout ("wave chunks", replaces swarm/types/ensemble.hpp:131-150,227-238 +
 * types/coalescedstructarray.hpp:135-140 with CHUNK 8 -> 64):
 *   chunk k holds systems 64k .. 64k+63, one system per lane of a wavefront.
 *   body : double [nchunk][nbod][7 + nbattr][64]   fields x,y,z,vx,vy,vz,mass,attr0..
 *   sysd : double [nchunk][1 + nsattr][64]         fields time,attr0..
 *   sysi : int32  [nchunk][2][64]                  fields state,id
 *   nchunk = ceil(nsys / 64).  Lanes >= nsys in the last chunk are padding and are ignored.
 */
#ifndef SWARM_HIP_H
#define SWARM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWARM_HIP_CHUNK 64
#define SWARM_HIP_MIN_NBOD 3
/* nbod 3 .. 10 run one lane per system or one lane per body (swarm_hip_launch_t::lane_mode); 11 .. 32 one lane per
 * body only.  The reference instantiates 3 .. MAX_NBODIES, accepts MAX_NBODIES up to 32 (cmake/variables.cmake:55-72,
 * default 6) and ships golden outputs up to nbod 13 (test/integrators/test.13.out.txt). */
#define SWARM_HIP_MAX_NBOD 32
#define SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM 10

/* error codes */
#define SWARM_HIP_OK 0
#define SWARM_HIP_EINVAL (-1)     /* bad argument (nbod out of range, NULL pointer, ...) */
#define SWARM_HIP_EHIP (-2)       /* a HIP runtime call failed */
#define SWARM_HIP_EUNSUPPORTED (-3)

/* system states, swarm/types/ensemble.hpp:269-274 */
#define SWARM_HIP_ACTIVE 0
#define SWARM_HIP_INACTIVE 1
#define SWARM_HIP_NEEDS_EXAM 2
#define SWARM_HIP_DISABLED (-1)

/* integrator ids = the reference's registry names (src/plugins/ *.cu) */
enum swarm_hip_integrator {
    SWARM_HIP_HERMITE = 1,       /* plugins/hermite.cu:40, integrators/hermite_integrator.hpp */
    SWARM_HIP_HERMITE_ADAP = 2,  /* plugins/hermite_adap.cu:39-47, integrators/hermite_adap.hpp */
    SWARM_HIP_RKCK_FIXED = 3,    /* plugins/rkck_fixed.cu:28, integrators/rkck_integrator.hpp */
    SWARM_HIP_RKCK_ADAPTIVE = 4, /* plugins/rkck_adaptive.cu:27 */
    SWARM_HIP_MVS = 5,           /* plugins/mvs.cu:37, propagators/mvs.hpp + keplerian.hpp */
    SWARM_HIP_EULER = 6,         /* plugins/euler.cu:36, propagators/euler.hpp */
    SWARM_HIP_VERLET = 7,        /* plugins/verlet.cu:35, propagators/verlet.hpp */
    SWARM_HIP_MIDPOINT = 8,      /* plugins/midpoint.cu:35, propagators/midpoint.hpp */
    SWARM_HIP_HERMITE_PROP = 9,  /* plugins/hermite_prop.cu:37, propagators/hermite_propagator.hpp */
    SWARM_HIP_IRK2 = 10,         /* plugins/irk2.cu:39, integrators/irk2_integrator.hpp (Gauss IRK) */
    SWARM_HIP_IRK2_CPU = 11      /* plugins/irk2_cpu.cpp:36, integrators/irk2_cpu.hpp: same kernel with
                                    irk2_cpu's constants (s = 6, uround 1e-16, 50 sweeps, failure rule) */
};

/* monitor bound to the plugin name */
enum swarm_hip_monitor {
    SWARM_HIP_MON_EJECTION = 0,          /* monitors/stop_on_ejection.hpp */
    SWARM_HIP_MON_LOG_INTERVAL = 1,      /* monitors/log_time_interval.hpp (`*_log` names) */
    SWARM_HIP_MON_EJECTION_OR_CLOSE = 2, /* monitors/composites.hpp:145-222 (`*_close_encounter`) */
    SWARM_HIP_MON_MCE_STAT = 3,          /* monitors/mercury_mce_stat.hpp: what plugins/hermite.cu:40 binds
                                            to `hermite`; SWARM_HIP_HERMITE only, needs nbattr >= 3
                                            (radius, rce, rphysics) and nsattr >= 1 */
    SWARM_HIP_MON_LOG_TRANSIT = 4       /* monitors/log_transit.hpp (`hermite_adap_transit`,
                                            plugins/hermite_adap.cu:61); SWARM_HIP_HERMITE_ADAP only */
};

typedef struct swarm_hip_ensemble {
    int32_t nsys, nbod;
    int32_t nbattr, nsattr; /* NUM_PLANET_ATTRIBUTES / NUM_SYSTEM_ATTRIBUTES, src/config.h.in */
    double *body;
    double *sysd;
    int32_t *sysi;
} swarm_hip_ensemble_t;

/* Device event buffer: replaces gpulog::log_base {char* buffer; int* at; int buf_len}
 * (swarm/log/gpulog/bits/gpulog_log.h:171-178).  Records are byte-compatible with the
 * reference (EVT_SNAPSHOT layout: log/log.hpp:313-326, log/types.hpp:37-56).
 * d_cursor[0] = bytes used, d_cursor[2] = lprintf (MSG_PRINTF) records among them (so that a flush
 * only walks the buffer for replay_printf when there is something to print), d_cursor[3] reserved,
 * d_cursor[1] = records dropped because the buffer was full
 * (the reference drops them silently, bits/gpulog_write.h:14-22).
 * Read or clear the buffer and its cursor only after the launch stream has been synchronised: a flush that
 * overlaps a running kernel loses the records appended after the copy. */
typedef struct swarm_hip_log {
    uint8_t *d_buf;
    int32_t *d_cursor; /* int32[4], zeroed by the caller */
    int32_t capacity;  /* bytes */
} swarm_hip_log_t;

/* One kernel launch of an integrator plugin: what `launch_integrator()` ->
 * launch_templatized_integrator (swarm/gpu/helpers.hpp:167-238) does in the reference.
 * Config keys are the reference's (.cfg names in comments). */
typedef struct swarm_hip_launch {
    int32_t integrator;     /* enum swarm_hip_integrator          integrator=            */
    int32_t monitor;        /* enum swarm_hip_monitor             (by plugin name)       */
    int32_t max_iterations; /* steps per launch per system        max_iterations=        */
    int32_t block_size;     /* threads per workgroup, multiple of 64, at most 256; 0 = default
                               (256); the analogue of system_per_block=.  64 and 256 measure the
                               same; 128 and 192 are accepted but ran up to 2x slower where the
                               ensemble gives every SIMD exactly one wavefront (65536 systems:
                               hermite nbod 3 8.4 -> 16.2 ms), so the hosts map to 64 or 256     */
    double destination_time; /*                                   destination_time=      */
    double time_step;        /* hermite(_prop), mvs, euler, midpoint, irk2 time_step=    */
    double time_step_factor; /* hermite_adap                      time_step_factor=      */
    double min_time_step;    /* hermite_adap, rkck                min_time_step=         */
    double max_time_step;    /* rkck                              max_time_step=         */
    double error_tolerance;  /* rkck                              error_tolerance=       */
    double max_timestep;     /* verlet                            max_timestep=          */
    double timestep_scale;   /* verlet                            timestep_scale=        */
    /* monitors */
    double rmax;                           /* rmax= (default FLT_MAX)                    */
    int32_t deactivate_on_ejection;        /* deactivate_on_ejection=                    */
    int32_t log_on_ejection;               /* log_on_ejection=                           */
    int32_t verbose_on_ejection;           /* verbose_on_ejection= (lprintf records)     */
    int32_t deactivate_on_close_encounter; /* deactivate_on_close_encounter=             */
    int32_t log_on_close_encounter;        /* log_on_close_encounter=                    */
    int32_t verbose_on_close_encounter;    /* verbose_on_close_encounter=                */
    double close_approach;                 /* close_approach=                            */
    double log_interval;                   /* log_interval=                              */
    double rceh;                           /* rceh= (mce_stat, default 1e-6)             */
    int32_t irk2_method;                   /* method= (irk2: stages 2, 4, 6; 0 = 6)      */
    int32_t log_transits;                  /* log_transits= (log_transit, default 1)     */
    int32_t log_occultations;              /* log_occultations= (default 0)              */
    int32_t lane_mode;                     /* 0 = choose by nbod and nsys; 1 = one lane per system; 2 = one lane
                                              per body (hermite, hermite_adap, mvs, rkck_*, euler, midpoint,
                                              verlet, hermite_prop: small ensembles;
                                              irk2 from nbod 4 and rkck_* from nbod 9 on at every size; each
                                              with a no-op monitor, stop_on_ejection, log_time_interval or the
                                              ejection-or-close-encounter composite without printing; ignored
                                              elsewhere).  No reference key. */
    double log_transit_tol;                /* log_transit_tol= (default 2e-8; 0 = off); the
                                              monitor's dt is time_step=                 */
    int32_t kepler_compensated;            /* kepler_compensated= (mvs; default 0; no reference key).  0: the f and g
                                              functions of the Kepler drift exactly as propagators/keplerian.hpp:214-224
                                              evaluates them, `dfdt = (f*dgdt - 1)/g` with f and dgdt rounded to doubles
                                              next to 1 -- a deterministic error of ~1e-16/time_step per step.  1: the same
                                              expressions evaluated from f-1 and dgdt-1 (no cancellation); DESIGN.md 4 */
    int32_t active_hint;                   /* number of ACTIVE systems if the caller knows it -- the result of its
                                              swarm_hip_count_active after the previous launch of the attempt loop
                                              (swarm/integrator.cpp:102-117) -- else 0.  When fewer systems are ACTIVE
                                              than the ensemble holds, the choice between the two kernel layouts goes by
                                              this number, and a lane-per-body launch first gathers the ACTIVE systems
                                              into dense wavefronts (on the device, no synchronisation).  Results do
                                              not depend on it; a hint below the true count leaves the surplus systems
                                              ACTIVE for the next launch.  No reference key. */
    swarm_hip_log_t log;                   /* d_buf may be NULL: records are dropped     */
    /* optional per-system step counters, uint64 [nchunk*64] each, accumulated (+=) */
    uint64_t *d_steps_accepted;
    uint64_t *d_steps_attempted;
    void *stream;
} swarm_hip_launch_t;

/* ---- library ----------------------------------------------------------------------------- */
/* Select `device` (hipSetDevice) and create the per-device scratch the helpers use.
 * Replaces select_cuda_device (swarm/gpu/device_settings.cpp:52-62).  Synchronous. */
int swarm_hip_init(int device);
const char *swarm_hip_last_error(void);
const char *swarm_hip_version(void);

/* ---- layout helpers (host, no GPU touched) ------------------------------------------------ */
/* Element counts, cf. ensemble::body_element_count / sys_element_count (ensemble.hpp:448-455) */
int64_t swarm_hip_nchunk(int32_t nsys);
size_t swarm_hip_body_bytes(int32_t nsys, int32_t nbod, int32_t nbattr);
size_t swarm_hip_sysd_bytes(int32_t nsys, int32_t nsattr);
size_t swarm_hip_sysi_bytes(int32_t nsys);
int64_t swarm_hip_snapshot_record_bytes(int32_t nbod); /* 32 + 64*nbod rounded to 16 */

/* ---- the hot path ------------------------------------------------------------------------ */
/* Integrate every ACTIVE system until destination_time or max_iterations steps, with the
 * monitor fused.  = generic_kernel<Impl,T><<<>>> (swarm/gpu/helpers.hpp:124-127,198). */
int swarm_hip_launch_integrator(const swarm_hip_ensemble_t *ens, const swarm_hip_launch_t *args);

/* Count systems with state == ACTIVE.  = number_of_active_systems (swarm/gpu/utilities.cu:30-64).
 * Synchronises `stream`; result in *count. */
int swarm_hip_count_active(const swarm_hip_ensemble_t *ens, void *stream, int32_t *count);

/* The same count for callers that cannot allocate or synchronise (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Asynchronous on
 * `stream`; the result goes to the caller's DEVICE int32 *d_count (written, not accumulated: it need not
 * be cleared).  Calls no hipMalloc/hipFree: the per-workgroup partial counts live in a static device array
 * of 64 KiB that every GPU the library runs on holds until the process ends (one int32 per 4096 systems),
 * hence the limit nsys <= SWARM_HIP_NOALLOC_MAX_NSYS (larger: SWARM_HIP_EINVAL).  All calls on one GPU
 * share that array: calls that overlap on one GPU (two streams, two host threads) race -- order them.
 * swarm_hip_init(device) must have run for the current device (it resolves the array's address there, once);
 * without it the call returns SWARM_HIP_EINVAL.  The call itself takes no lock. */
#define SWARM_HIP_NOALLOC_MAX_NSYS (1 << 26)
int swarm_hip_count_active_noalloc(const swarm_hip_ensemble_t *ens, int32_t *d_count, void *stream);

/* INACTIVE -> ACTIVE.  = reactivate_systems (swarm/gpu/utilities.cu:113-131) and
 * activate_inactive_systems (swarm/integrator.cpp:83-89). */
int swarm_hip_reactivate(const swarm_hip_ensemble_t *ens, void *stream);

/* Total energy per system into d_energy[nchunk*64] (K + U, G = 1).
 * = SystemRef::calc_total_energy (swarm/types/ensemble.hpp:402-413). */
int swarm_hip_total_energy(const swarm_hip_ensemble_t *ens, double *d_energy, void *stream);

/* Reduction over systems for the multi-GPU statistics (no counterpart kernel in the reference,
 * which reduces on the host: energy_conservation_error_range, swarm/utils.cpp:108-121).
 * Given d_e0/d_e1 [nsys] writes d_out[8] = {sum |dE/E0|, max |dE/E0|, min |dE/E0|, n_active,
 * n_inactive, n_disabled, n_other, nsys}. */
int swarm_hip_energy_stats(const swarm_hip_ensemble_t *ens, const double *d_e0, const double *d_e1,
                           double *d_out, void *stream);

/* ---- layout conversion ------------------------------------------------------------------- */
/* Reference CoalescedStructArray image (device copies of ensemble::Body[] and ensemble::Sys[]
 * built with ENSEMBLE_CHUNK_SIZE = ref_chunk; swarm/types/ensemble.hpp:131-150,227-238,499-517)
 * -> wave chunks, and back.  This is what a reference-side `gpu::integrator` calls after
 * upload_ensemble / before download_ensemble (swarm/integrator.hpp:266-273). */
int swarm_hip_import_ref_layout(const swarm_hip_ensemble_t *ens, const void *d_ref_body,
                                const void *d_ref_sys, int32_t ref_chunk, void *stream);
int swarm_hip_export_ref_layout(const swarm_hip_ensemble_t *ens, void *d_ref_body, void *d_ref_sys,
                                int32_t ref_chunk, void *stream);

/* Plain host AoS arrays <-> device wave chunks (synchronous; the arrays are copied as they are into a
 * device staging buffer and transposed on the GPU.  Each device keeps its staging buffer between calls,
 * grow-only, for ensembles up to 1 GiB; calls that overlap on one GPU are serialised).
 * pos/vel [nsys][nbod][3], mass [nsys][nbod], time [nsys], state/id [nsys],
 * body_attr [nsys][nbod][nbattr] (may be NULL), sys_attr [nsys][nsattr] (may be NULL).
 * = alloc_copy H2D/D2H (swarm/types/allocators.hpp:152-166) for callers without the
 * reference's host ensemble class. */
int swarm_hip_upload_aos(const swarm_hip_ensemble_t *ens, const double *pos, const double *vel,
                         const double *mass, const double *time, const int32_t *state,
                         const int32_t *id, const double *body_attr, const double *sys_attr);
int swarm_hip_download_aos(const swarm_hip_ensemble_t *ens, double *pos, double *vel, double *mass,
                           double *time, int32_t *state, int32_t *id, double *body_attr,
                           double *sys_attr);

/* ---- device memory convenience (hipMalloc/hipFree/hipMemcpy wrappers for FFI callers) ------ */
int swarm_hip_malloc(void **d_ptr, size_t bytes);
int swarm_hip_free(void *d_ptr);
int swarm_hip_memset(void *d_ptr, int value, size_t bytes, void *stream);
int swarm_hip_memcpy_d2h(void *host, const void *d_ptr, size_t bytes);
int swarm_hip_memcpy_h2d(void *d_ptr, const void *host, size_t bytes);
int swarm_hip_synchronize(void *stream);
/* page-locked host memory for the host side of the event log (gpulog's host buffer is pinned
 * too: hostAlloc in swarm/log/gpulog/bits/gpulog_log.h); D2H copies into it run at PCIe speed */
int swarm_hip_host_alloc(void **host_ptr, size_t bytes);
int swarm_hip_host_free(void *host_ptr);
/* timing on the launch stream (HIP events): elapsed ms between two marks */
int swarm_hip_event_create(void **event);
int swarm_hip_event_record(void *event, void *stream);
int swarm_hip_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises stop */
int swarm_hip_event_destroy(void *event);
int swarm_hip_event_synchronize(void *event);
/* ---- overlapping the log flush with the next launch ---------------------------------------- */
/* The reference flushes the device log after every launch with a synchronous copy
 * (gpulog::copy, swarm/log/gpulog/bits/gpulog_log.h:343-394, called by log::manager::flush,
 * swarm/log/logmanager.cpp:67-87).  With two device logs a host can do the same protocol while the
 * next launch runs: launch k fills log A; an event recorded behind it; a second stream waits for
 * that event and copies A's used bytes into page-locked memory while launch k+1 fills log B.
 * These four calls are all it needs: a stream that is not ordered against the NULL stream,
 * stream-waits-for-event, an asynchronous D2H copy (`host` must be page-locked to overlap), and
 * (above) swarm_hip_event_synchronize. */
int swarm_hip_stream_create(void **stream);
int swarm_hip_stream_destroy(void *stream);
int swarm_hip_stream_wait_event(void *stream, void *event);
int swarm_hip_memcpy_d2h_async(void *host, const void *d_ptr, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
