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
    hipStream_t stream;
};

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
        set_error("nbod=%d outside [%d,%d]", nbod, SWARM_HIP_MIN_NBOD, SWARM_HIP_MAX_NBOD); \
        return SWARM_HIP_EINVAL;                                           \
    }
#endif

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SWARM_HIP_EHIP;
    }
    return SWARM_HIP_OK;
}

} // namespace swh
