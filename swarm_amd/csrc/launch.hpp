// launch.hpp -- host-side glue shared by the kernel translation units of libswarm_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.hpp"

namespace swh {

struct LaunchCtx {
    EnsView ens;
    IntParams ip;
    MonParams mp;
    LogBuf lg;
    Counters cnt;
    int block;      // threads per workgroup
    int grid;       // workgroups
    int lane_mode;  // swarm_hip_launch_t::lane_mode
    int active_hint; // swarm_hip_launch_t::active_hint: ACTIVE systems as last counted by the caller, 0 = not known
    hipStream_t stream;
    // the number of systems that the choice between the two kernel layouts goes by: the ACTIVE ones where known
    int effective_nsys() const { return (active_hint > 0 && active_hint < ens.nsys) ? active_hint : ens.nsys; }
};

// Lane-per-body launches (capi.hip).  prepare: `view` = the ensemble as the kernel sees it and `grid` = its workgroups
// (one wavefront of G systems each).  When the caller's hint says that fewer systems are ACTIVE than the ensemble
// holds, the ACTIVE ones are first gathered into a list on the device (stream-ordered allocation, one small kernel) and
// the kernel walks that list: dense wavefronts for the tail of an adaptive or monitored run.  `premonitor`: the
// integrator calls its monitor once before the step loop for EVERY system (hermite_adap, rkck) -- the systems that are
// not ACTIVE then get a launch of their own over `rest` (rest_grid workgroups; they take no steps), so that states
// and log records stay what the ungathered launch produces.  finish: gives the lists back (stream-ordered), after
// the kernels have been enqueued.
struct BodyLaunch {
    EnsView view, rest;
    int grid, rest_grid;
    int32_t *mem;
};
int prepare_body_launch(const LaunchCtx &c, int G, bool premonitor, BodyLaunch &bl);
void finish_body_launch(const LaunchCtx &c, BodyLaunch &bl);

void set_error(const char *fmt, ...);

// one entry per kernel translation unit; each dispatches on nbod
int launch_hermite(const LaunchCtx &c, bool adaptive);
int launch_hermite_body(const LaunchCtx &c, bool adaptive); // one lane per body, small ensembles (kernels_hermite_body.hip)
int launch_hermite_mce(const LaunchCtx &c); // `hermite` + mce_stat (plugins/hermite.cu:40)
int launch_rkck(const LaunchCtx &c, bool adaptive);
int launch_rkck_body(const LaunchCtx &c, bool adaptive); // one lane per body, small ensembles (kernels_rkck_body.hip)
int launch_generic(const LaunchCtx &c, int integrator);
int launch_prop_body(const LaunchCtx &c, int integrator); // euler, midpoint, hermite_prop: one lane per body
int launch_mvs_body(const LaunchCtx &c); // one lane per body, small ensembles (kernels_mvs_body.hip)
// Gauss IRK; method = stages (2, 4, 6); cpu_rules = irk2_cpu's iteration constants and failure rule
int launch_irk2(const LaunchCtx &c, int method, bool cpu_rules);
int launch_irk2_s2(const LaunchCtx &c, bool cpu_rules); // one object per stage count, kernels_irk2.hip
int launch_irk2_s4(const LaunchCtx &c, bool cpu_rules);
int launch_irk2_s6(const LaunchCtx &c, bool cpu_rules);
int launch_irk2_body(const LaunchCtx &c, int method, bool cpu_rules); // one lane per body (kernels_irk2_body.hip)

// nbod -> template dispatch, the analogue of choose<> (swarm/gpu/helpers.hpp:97-107,216-238)
#ifdef SWH_DISPATCH_ONLY /* kernel experiments: instantiate one nbod only (-DSWH_DISPATCH_ONLY=3), never in the shipped build */
#define SWH_DISPATCH_NBOD(nbod, CALL)                                      \
    if (nbod == SWH_DISPATCH_ONLY) { constexpr int N = SWH_DISPATCH_ONLY; CALL; } \
    else { set_error("experiment build: nbod=%d only", SWH_DISPATCH_ONLY); return SWARM_HIP_EINVAL; }
#define SWH_DISPATCH_NBOD_BODY(nbod, CALL) SWH_DISPATCH_NBOD(nbod, CALL)
#else
#define SWH_DISPATCH_NBOD(nbod, CALL)                                      \
    switch (nbod) {                                                        \
    case 3: { constexpr int N = 3; CALL; } break;                          \
    case 4: { constexpr int N = 4; CALL; } break;                          \
    case 5: { constexpr int N = 5; CALL; } break;                          \
    case 6: { constexpr int N = 6; CALL; } break;                          \
    case 7: { constexpr int N = 7; CALL; } break;                          \
    case 8: { constexpr int N = 8; CALL; } break;                          \
    case 9: { constexpr int N = 9; CALL; } break;                          \
    case 10: { constexpr int N = 10; CALL; } break;                        \
    default:                                                               \
        set_error("nbod=%d outside [%d,%d] (one lane per system)", nbod, SWARM_HIP_MIN_NBOD, SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM); \
        return SWARM_HIP_EUNSUPPORTED;                                     \
    }
// the lane-per-body kernels keep one body per lane whatever nbod is: 3 .. SWARM_HIP_MAX_NBOD (floor(64/nbod) systems
// per wavefront); the reference's ctest runs nbod = 3 .. MAX_NBODIES and ships goldens up to 13
#define SWH_DISPATCH_NBOD_BODY(nbod, CALL)                                 \
    switch (nbod) {                                                        \
    case 3: { constexpr int N = 3; CALL; } break;                          \
    case 4: { constexpr int N = 4; CALL; } break;                          \
    case 5: { constexpr int N = 5; CALL; } break;                          \
    case 6: { constexpr int N = 6; CALL; } break;                          \
    case 7: { constexpr int N = 7; CALL; } break;                          \
    case 8: { constexpr int N = 8; CALL; } break;                          \
    case 9: { constexpr int N = 9; CALL; } break;                          \
    case 10: { constexpr int N = 10; CALL; } break;                        \
    case 11: { constexpr int N = 11; CALL; } break;                        \
    case 12: { constexpr int N = 12; CALL; } break;                        \
    case 13: { constexpr int N = 13; CALL; } break;                        \
    case 14: { constexpr int N = 14; CALL; } break;                        \
    case 15: { constexpr int N = 15; CALL; } break;                        \
    case 16: { constexpr int N = 16; CALL; } break;                        \
    case 17: { constexpr int N = 17; CALL; } break;                        \
    case 18: { constexpr int N = 18; CALL; } break;                        \
    case 19: { constexpr int N = 19; CALL; } break;                        \
    case 20: { constexpr int N = 20; CALL; } break;                        \
    case 21: { constexpr int N = 21; CALL; } break;                        \
    case 22: { constexpr int N = 22; CALL; } break;                        \
    case 23: { constexpr int N = 23; CALL; } break;                        \
    case 24: { constexpr int N = 24; CALL; } break;                        \
    case 25: { constexpr int N = 25; CALL; } break;                        \
    case 26: { constexpr int N = 26; CALL; } break;                        \
    case 27: { constexpr int N = 27; CALL; } break;                        \
    case 28: { constexpr int N = 28; CALL; } break;                        \
    case 29: { constexpr int N = 29; CALL; } break;                        \
    case 30: { constexpr int N = 30; CALL; } break;                        \
    case 31: { constexpr int N = 31; CALL; } break;                        \
    case 32: { constexpr int N = 32; CALL; } break;                        \
    default:                                                               \
        set_error("nbod=%d outside [%d,%d]", nbod, SWARM_HIP_MIN_NBOD, SWARM_HIP_MAX_NBOD); \
        return SWARM_HIP_EINVAL;                                           \
    }
#endif

// nbod beyond the lane-per-system kernels' range runs one lane per body whatever the ensemble size
inline bool body_only_nbod(const LaunchCtx &c) { return c.ens.nbod > SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM; }

// The layout a launch takes: 1 = one lane per body, 0 = one lane per system, < 0 = error (set).  `body_ok`: the
// lane-per-body kernel serves the bound monitor; `prefers_body`: the integrator's measured crossover says so for this
// many (ACTIVE) systems.  lane_mode 1 / 2 force a layout where both exist.
inline int choose_layout(const LaunchCtx &c, bool body_ok, bool prefers_body) {
    if (body_only_nbod(c)) {
        if (body_ok) return 1;
        set_error("nbod=%d runs one lane per body (nbod > %d), which serves the no-op monitor, stop_on_ejection, "
                  "log_time_interval and the ejection-or-close-encounter composite; mce_stat and log_transit stop at "
                  "nbod %d",
                  c.ens.nbod, SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM, SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM);
        return SWARM_HIP_EUNSUPPORTED;
    }
    return (body_ok && (c.lane_mode == 2 || (c.lane_mode == 0 && prefers_body))) ? 1 : 0;
}
// crossover tables hold one limit per nbod 0 .. SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM
inline bool within_limit(const LaunchCtx &c, const int (&limit)[SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM + 1]) {
    return c.ens.nbod <= SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM && c.effective_nsys() <= limit[c.ens.nbod];
}

// lane-per-body kernels built in both loop forms (BodyLoops, device_common.hpp): `from` holds, per nbod up to
// BODY_UNROLL_MAX, the number of (ACTIVE) systems from which the rolled form is ahead, 0 = never
inline bool rolled_form(const LaunchCtx &c, const int (&from)[BODY_UNROLL_MAX + 1]) {
    const int n = c.ens.nbod;
    return n > BODY_UNROLL_MAX || (from[n] > 0 && c.effective_nsys() >= from[n]);
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SWARM_HIP_EHIP;
    }
    return SWARM_HIP_OK;
}

} // namespace swh
