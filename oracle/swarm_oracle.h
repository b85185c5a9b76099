/*
 * swarm_oracle.h -- CPU restatement of the Swarm-NG ensemble-integrator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under swarm_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline, never as the product path.
 *
 * Pinned against the reference's own golden vectors (tests/golden/, copied data files of
 * /root/reference/test/): see tests/test_oracle_golden.py.  The reference itself is not
 * buildable in this image without stand-in headers (boost, CUDA runtime, generated config.h),
 * so no oracle/_ref build exists; see DESIGN.md "Oracle".
 *
 * Host layout (plain AoS, independent of the device layout):
 *   pos[nsys][nbod][3], vel[nsys][nbod][3], mass[nsys][nbod], time[nsys], state[nsys], id[nsys]
 */
#ifndef SWARM_ORACLE_H
#define SWARM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NBOD 32

/* reference: src/swarm/types/ensemble.hpp:269-274 */
enum { ORC_ACTIVE = 0, ORC_INACTIVE = 1, ORC_NEEDS_EXAM = 2, ORC_DISABLED = -1 };

typedef struct {
    int32_t nsys, nbod;
    double *pos, *vel, *mass, *time;
    int32_t *state, *id;
    /* attributes (ensemble.hpp:131-150,227-238): battr[nsys][nbod][nbattr], sattr[nsys][nsattr];
     * may be NULL with nbattr/nsattr 0 for integrators/monitors that never touch them */
    int32_t nbattr, nsattr;
    double *battr, *sattr;
} orc_ensemble;

/* monitor kinds: which reference monitor class the plugin name binds (src/plugins/ *.cu) */
enum { ORC_MON_EJECTION = 0, ORC_MON_LOG_INTERVAL = 1, ORC_MON_EJECTION_OR_CLOSE = 2,
       ORC_MON_MCE_STAT = 3, /* monitors/mercury_mce_stat.hpp, bound to `hermite` (plugins/hermite.cu:40) */
       ORC_MON_LOG_TRANSIT = 4 /* monitors/log_transit.hpp, `hermite_adap_transit` (plugins/hermite_adap.cu:61) */ };

typedef struct {
    int32_t kind;
    /* stop_on_ejection_params, src/monitors/stop_on_ejection.hpp:42-54 */
    double rmax;
    int32_t deactivate_on_ejection, log_on_ejection, verbose_on_ejection;
    /* log_time_interval_params, src/monitors/log_time_interval.hpp:35-44 */
    double log_interval;
    /* stop_on_close_encounter_param, src/monitors/stop_on_close_encounter.hpp */
    double close_approach;
    int32_t deactivate_on_close_encounter, log_on_close_encounter, verbose_on_close_encounter;
    /* mce_stat_params, src/monitors/mercury_mce_stat.hpp:36-55 (dmin = close_approach, unused) */
    double rceh;          /* rceh= (default 1e-6) */
    double mce_time_step; /* time_step= */
    /* log_transit_params, src/monitors/log_transit.hpp:42-58 (dt = time_step=) */
    double tr_tol, tr_dt;
    int32_t tr_log_transit, tr_log_occultation;
} orc_monitor;

/* append-only record buffer: the host twin of gpulog::log_base (bits/gpulog_log.h:171-178) */
typedef struct {
    uint8_t *buf;
    int64_t cap;
    int64_t len;      /* bytes used */
    int64_t dropped;  /* records that did not fit (gpulog_write.h:17-21 returns NULL) */
} orc_log;

/* integrator ids (shared numbering with include/swarm_hip.h) */
enum {
    ORC_INT_HERMITE_CPU = 0, /* integrators/hermite_cpu.hpp (CPU arithmetic order)        */
    ORC_INT_HERMITE = 1,     /* integrators/hermite_integrator.hpp (GPU arithmetic order) */
    ORC_INT_HERMITE_ADAP = 2,
    ORC_INT_RKCK_FIXED = 3,
    ORC_INT_RKCK_ADAPTIVE = 4,
    ORC_INT_MVS = 5,
    ORC_INT_EULER = 6,
    ORC_INT_VERLET = 7,
    ORC_INT_MIDPOINT = 8,
    ORC_INT_HERMITE_PROP = 9, /* propagators/hermite_propagator.hpp under generic<> */
    ORC_INT_IRK2_CPU = 10,    /* integrators/irk2_cpu.hpp (CPU arithmetic order, s = 6)     */
    ORC_INT_IRK2 = 11         /* integrators/irk2_integrator.hpp (GPU arithmetic order)     */
};

typedef struct {
    int32_t integrator;
    int32_t max_iterations;   /* per launch; rkck overrides with dest/max_time_step */
    double destination_time;
    double time_step;         /* hermite, mvs, euler, midpoint */
    double time_step_factor;  /* hermite_adap */
    double min_time_step;     /* hermite_adap, rkck */
    double max_time_step;     /* rkck */
    double error_tolerance;   /* rkck */
    double max_timestep;      /* verlet: max_timestep */
    double timestep_scale;    /* verlet */
    int32_t irk2_method;      /* irk2: method= (number of stages 2, 4 or 6; 0 = default 6) */
    int32_t kepler_compensated; /* mvs: 0 = the f, g update exactly as propagators/keplerian.hpp:214-231 writes it;
                                   1 = the same expressions from f-1 and dg/dt-1 (no reference counterpart: the
                                   restatement of the product's kepler_compensated=1, used to show that everything
                                   else in the MVS restatement agrees with the hermite_cpu goldens to 1e-13) */
} orc_params;

/* per-system counters, optional (may be NULL) */
typedef struct {
    int64_t *accepted;   /* [nsys] accepted integrator steps, accumulated */
    int64_t *attempted;  /* [nsys] attempted steps (rkck rejects count here) */
} orc_counters;

void orc_generate(const orc_ensemble *e, double spacing_factor, double planet_mass,
                  double ejection_factor);
/* one "kernel launch" of the named integrator over every system; threads>1 uses OpenMP */
int orc_launch(const orc_ensemble *e, const orc_params *p, const orc_monitor *m, orc_log *log,
               const orc_counters *cnt, int threads);
/* the reference's host attempt loop: reactivate, launch until no system is active */
int orc_integrate(const orc_ensemble *e, const orc_params *p, const orc_monitor *m, orc_log *log,
                  const orc_counters *cnt, int threads, int max_attempts);

void orc_total_energy(const orc_ensemble *e, double *energy);
int orc_compare(const orc_ensemble *a, const orc_ensemble *b, double *pos_diff, double *vel_diff,
                double *time_diff);
int orc_count_active(const orc_ensemble *e);
int orc_pair_first(int nbod, int ij);
int orc_pair_second(int nbod, int ij);
int64_t orc_snapshot_record_size(int nbod);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
