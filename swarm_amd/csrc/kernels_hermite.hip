// kernels_hermite.hip -- fused PEC2 Hermite kernels (fixed and adaptive step) for gfx950.
//
// Replaces hermite<Monitor,Grav>::kernel (integrators/hermite_integrator.hpp:74-187) and
// hermite_adap<Monitor,Grav>::kernel (integrators/hermite_adap.hpp:68-201) of the reference:
// predictor, two force evaluations + correctors, time update, monitor and end-of-integration
// test are one loop body over register-resident state; K steps per launch.
#include "launch.hpp"

namespace swh {

template <int N, bool ADAPTIVE, bool MON>
__global__ __launch_bounds__(256) void hermite_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
                                                      Counters cnt) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    State<N> s;
    load_state(e, gid, s);

    const int state_in = s.state;
    MonState ms;
    mon_init(ms, s);
    // hermite_adap.hpp:130: the adaptive kernel calls the monitor once before the loop,
    // for every system of the ensemble whatever its state.
    if (ADAPTIVE && MON) mon_call(ms, s, mp, lg, true);

    double a0[N][3], j0[N][3];
    unsigned long long steps = 0;
    if (s.state == SWARM_HIP_ACTIVE) {
        accjerk<N>(s.p, s.v, s.m, a0, j0);

        const double dest = ip.destination_time;
        for (int iter = 0; iter < ip.max_iterations && s.state == SWARM_HIP_ACTIVE; iter++) {
            double h;
            if (ADAPTIVE) {
                // calc_adaptive_time_step, hermite_adap.hpp:69-98: sum over bodies of |jerk|^2/|acc|^2
                double tf = 0.0;
#pragma unroll
                for (int b = 0; b < N; b++) {
                    const double a2 = a0[b][0] * a0[b][0] + a0[b][1] * a0[b][1] + a0[b][2] * a0[b][2];
                    const double jj = j0[b][0] * j0[b][0] + j0[b][1] * j0[b][1] + j0[b][2] * j0[b][2];
                    const double ratio = jj / a2;
                    tf = (b == 0) ? ratio : tf + ratio;
                }
                h = rsqrt_full(tf) * ip.time_step_factor + ip.min_time_step;
            } else {
                h = ip.time_step;
            }
            {
#pragma clang fp contract(off)
                if (s.t + h > dest) h = dest - s.t;
            }
            const double hh = h * 0.5, h3 = h / 3.0;
            const double h2 = h * h;
            const double c60 = (1.0 / 60.0) * h, c12 = (1.0 / 12.0) * h;

            // predict (hermite_integrator.hpp:131-132)
            double pp[N][3], pv[N][3];
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    pp[b][c] = s.p[b][c] + h * (s.v[b][c] + hh * (a0[b][c] + h3 * j0[b][c]));
                    pv[b][c] = s.v[b][c] + h * (a0[b][c] + hh * j0[b][c]);
                    s.p[b][c] = pp[b][c];
                    s.v[b][c] = pv[b][c];
                }

            // evaluate + correct, twice (hermite_integrator.hpp:137-162, factored form L146-147)
            double a1[N][3], j1[N][3];
#pragma unroll
            for (int round = 0; round < 2; round++) {
                accjerk<N>(s.p, s.v, s.m, a1, j1);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const double da = a0[b][c] - a1[b][c];
                        s.p[b][c] = pp[b][c] +
                                    ((0.1 - 0.25) * da - c60 * (7.0 * j0[b][c] + 2.0 * j1[b][c])) * h2;
                        s.v[b][c] = pv[b][c] + ((-0.5) * da - c12 * (5.0 * j0[b][c] + j1[b][c])) * h;
                    }
            }
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) a0[b][c] = a1[b][c], j0[b][c] = j1[b][c];

            s.t += h;
            steps++;
            if (MON) mon_call(ms, s, mp, lg, true);
            if (s.state == SWARM_HIP_ACTIVE) {
                // end test: hermite_integrator.hpp:177-179 (`>` dest-1e-12), hermite_adap.hpp:191-193 (`>=`)
#pragma clang fp contract(off)
                if (ADAPTIVE ? (s.t >= dest) : (s.t > dest - 1e-12)) s.state = SWARM_HIP_INACTIVE;
            }
        }
    }
    if (steps || s.state != state_in) store_state(e, gid, s);
    if (cnt.accepted) cnt.accepted[gid] += steps;
    if (cnt.attempted) cnt.attempted[gid] += steps;
}

template <int N, bool ADAPTIVE>
static int launch_n(const LaunchCtx &c) {
    if (monitor_is_noop(c.mp))
        hipLaunchKernelGGL((hermite_kernel<N, ADAPTIVE, false>), dim3(c.grid), dim3(c.block), 0, c.stream,
                           c.ens, c.ip, c.mp, c.lg, c.cnt);
    else
        hipLaunchKernelGGL((hermite_kernel<N, ADAPTIVE, true>), dim3(c.grid), dim3(c.block), 0, c.stream,
                           c.ens, c.ip, c.mp, c.lg, c.cnt);
    return check_launch("hermite kernel launch");
}

int launch_hermite(const LaunchCtx &c, bool adaptive) {
    if (adaptive) {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, true>(c)));
    } else {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, false>(c)));
    }
    return SWARM_HIP_OK;
}

} // namespace swh
