// kernels_hermite.hip -- fused PEC2 Hermite kernels (fixed and adaptive step) for gfx950.
//
// Replaces hermite<Monitor,Grav>::kernel (integrators/hermite_integrator.hpp:74-187) and
// hermite_adap<Monitor,Grav>::kernel (integrators/hermite_adap.hpp:68-201) of the reference:
// predictor, two force evaluations + correctors, time update, monitor and end-of-integration
// test are one loop body over register-resident state; K steps per launch.
#include "launch.hpp"

namespace swh {

// Per-step constants that depend only on h.  For the fixed-step integrator h changes at most once
// per launch (the clipped last step), so they are recomputed only when h changes.
struct StepConsts {
    double h, hh, h3;        // h, h*0.5, h/3.0             (predictor, hermite_integrator.hpp:131-132)
    double kp_a, kp_j0, kp_j1; // position corrector weights  (L146)
    double kv_a, kv_j0, kv_j1; // velocity corrector weights  (L147)
    __device__ __forceinline__ void set(double h_) {
        h = h_;
        hh = h_ * 0.5;
        h3 = h_ / 3.0;
        const double h2 = h_ * h_;
        const double c60 = (1.0 / 60.0) * h_, c12 = (1.0 / 12.0) * h_;
        // pos = pre + ((0.1-0.25)*(a0-a1) - 1/60*(7 j0 + 2 j1) h) h^2, expanded per operand
        kp_a = (0.1 - 0.25) * h2;
        kp_j0 = -(7.0 * c60) * h2;
        kp_j1 = -(2.0 * c60) * h2;
        // vel = pre + ((-0.5)*(a0-a1) - 1/12*(5 j0 + j1) h) h
        kv_a = (-0.5) * h_;
        kv_j0 = -(5.0 * c12) * h_;
        kv_j1 = -c12 * h_;
    }
};

// One PEC2 Hermite step on register-resident state: predictor, 2 x (force + corrector).
// (a0,j0) = forces at the start of the step, (a1,j1) receive the forces at the end; the caller
// alternates the two buffer pairs so no copy `acc0 = acc1` (hermite_integrator.hpp:163) is needed.
// The corrector is the reference's factored form expanded per operand: the terms in a0,j0 are
// summed once per step (c0,d0), each round adds the a1,j1 terms and makes ONE rounding at the
// magnitude of pos/vel, like the reference's final `pre_pos + (...)`.
template <int N>
__device__ __forceinline__ void hermite_step(State<N> &s, const StepConsts &k, const double (&a0)[N][3],
                                             const double (&j0)[N][3], double (&a1)[N][3],
                                             double (&j1)[N][3]) {
    double pp[N][3], pv[N][3], c0[N][3], d0[N][3];
#pragma unroll
    for (int b = 0; b < N; b++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            pp[b][c] = __builtin_fma(k.h, __builtin_fma(k.hh, __builtin_fma(k.h3, j0[b][c], a0[b][c]), s.v[b][c]),
                                     s.p[b][c]);
            pv[b][c] = __builtin_fma(k.h, __builtin_fma(k.hh, j0[b][c], a0[b][c]), s.v[b][c]);
            c0[b][c] = __builtin_fma(k.kp_a, a0[b][c], k.kp_j0 * j0[b][c]);
            d0[b][c] = __builtin_fma(k.kv_a, a0[b][c], k.kv_j0 * j0[b][c]);
            s.p[b][c] = pp[b][c];
            s.v[b][c] = pv[b][c];
        }
#pragma unroll
    for (int round = 0; round < 2; round++) {
        accjerk<N>(s.p, s.v, s.m, a1, j1);
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                s.p[b][c] = pp[b][c] + __builtin_fma(-k.kp_a, a1[b][c], __builtin_fma(k.kp_j1, j1[b][c], c0[b][c]));
                s.v[b][c] = pv[b][c] + __builtin_fma(-k.kv_a, a1[b][c], __builtin_fma(k.kv_j1, j1[b][c], d0[b][c]));
            }
    }
}

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

    unsigned long long steps = 0;
    if (s.state == SWARM_HIP_ACTIVE) {
        double fa[2][N][3], fj[2][N][3]; // ping-pong force buffers
        accjerk<N>(s.p, s.v, s.m, fa[0], fj[0]);
        const double dest = ip.destination_time;
        StepConsts k;
        k.set(ip.time_step);

        // everything of a step after h is known: step, time, monitor, end test
        auto finish_step = [&](const double h, const double (&a0)[N][3], const double (&j0)[N][3],
                               double (&a1)[N][3], double (&j1)[N][3]) {
            hermite_step<N>(s, k, a0, j0, a1, j1);
            s.t += h;
            steps++;
            if (MON) mon_call(ms, s, mp, lg, true);
            if (s.state == SWARM_HIP_ACTIVE) {
                // end test: hermite_integrator.hpp:177-179 (`>` dest-1e-12), hermite_adap.hpp:191-193 (`>=`)
#pragma clang fp contract(off)
                if (ADAPTIVE ? (s.t >= dest) : (s.t > dest - 1e-12)) s.state = SWARM_HIP_INACTIVE;
            }
        };

        int iter = 0;
        if (!ADAPTIVE) {
            // Fast path: full steps h = time_step with the constants fixed, two steps per trip
            // through ping-pong force buffers.  The clipped last step (t + h > dest) and odd
            // remainders fall through to the general loop below.
            const double dt = ip.time_step;
            auto full = [&]() {
#pragma clang fp contract(off)
                return iter < ip.max_iterations && s.state == SWARM_HIP_ACTIVE && !(s.t + dt > dest);
            };
            while (full()) {
                finish_step(dt, fa[0], fj[0], fa[1], fj[1]);
                iter++;
                if (full()) {
                    finish_step(dt, fa[1], fj[1], fa[0], fj[0]);
                    iter++;
                } else {
#pragma unroll
                    for (int b = 0; b < N; b++)
#pragma unroll
                        for (int c = 0; c < 3; c++) fa[0][b][c] = fa[1][b][c], fj[0][b][c] = fj[1][b][c];
                }
            }
        }
        // General loop: per-step h (adaptive step, or the clipped final step of the fixed-step form)
        while (iter < ip.max_iterations && s.state == SWARM_HIP_ACTIVE) {
            double h;
            if (ADAPTIVE) {
                // calc_adaptive_time_step, hermite_adap.hpp:69-98: sum over bodies of |jerk|^2/|acc|^2
                double tf = 0.0;
#pragma unroll
                for (int b = 0; b < N; b++) {
                    const double a2 = fa[0][b][0] * fa[0][b][0] + fa[0][b][1] * fa[0][b][1] + fa[0][b][2] * fa[0][b][2];
                    const double jj = fj[0][b][0] * fj[0][b][0] + fj[0][b][1] * fj[0][b][1] + fj[0][b][2] * fj[0][b][2];
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
            k.set(h);
            finish_step(h, fa[0], fj[0], fa[1], fj[1]);
            iter++;
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) fa[0][b][c] = fa[1][b][c], fj[0][b][c] = fj[1][b][c];
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
