// kernels_rkck.hip -- fused Cash-Karp Runge-Kutta kernels (fixed / adaptive) for gfx950.
//
// Replaces rkck<Adapt,Monitor,Grav>::kernel (integrators/rkck_integrator.hpp:90-294).  Step
// control is per lane: every system carries its own time_step, accepts or rejects on its own and
// retires on its own; the wavefront runs until its last system is done or the iteration budget
// (destination_time / max_time_step attempts, rkck_integrator.hpp:75,143) is used up.
//
// Deviation from the reference, on purpose: after a REJECTED attempt the reference leaves
// p6^2, v6^2 in the global ensemble (L221) until the next force evaluation overwrites them, so a
// launch whose budget ends on a reject hands back squares.  Here state is only written from the
// last ACCEPTED step.
#include "launch.hpp"

namespace swh {

template <int N, bool ADAPTIVE, bool MON>
__global__ __launch_bounds__(256) void rkck_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
                                                   Counters cnt) {
    // Cash-Karp constants, rkck_integrator.hpp:96-108
    const double b1 = 1.0 / 5.0;
    const double b2[] = {3.0 / 40.0, 9.0 / 40.0};
    const double b3[] = {0.3, -0.9, 1.2};
    const double b4[] = {-11.0 / 54.0, 2.5, -70.0 / 27.0, 35.0 / 27.0};
    const double b5[] = {1631.0 / 55296.0, 175.0 / 512.0, 575.0 / 13824.0, 44275.0 / 110592.0,
                         253.0 / 4096.0};
    const double b6[] = {37.0 / 378.0, 0, 250.0 / 621.0, 125.0 / 594.0, 0, 512.0 / 1771.0};
    const double ecc[] = {37.0 / 378.0 - 2825.0 / 27648.0, 0.0, 250.0 / 621.0 - 18575.0 / 48384.0,
                          125.0 / 594.0 - 13525.0 / 55296.0, -277.00 / 14336.0, 512.0 / 1771.0 - 0.25};

    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    State<N> s;
    load_state(e, gid, s);
    const int state_in = s.state;
    MonState ms;
    mon_init(ms, s);
    if (MON) mon_call(ms, s, mp, lg, true); // rkck_integrator.hpp:140, every system

    const int iteration_count = (int)(ip.destination_time / ip.max_time_step); // L75
    const double dest = ip.destination_time;
    double time_step = ip.max_time_step; // L133
    unsigned long long accepted = 0, attempted = 0;

    for (int iter = 0; iter < iteration_count && s.state == SWARM_HIP_ACTIVE; iter++) {
        double h = time_step;
        {
#pragma clang fp contract(off)
            if (s.t + h > dest) h = dest - s.t;
        }
        double kv[6][N][3], ka[6][N][3];
        double sp[N][3], sv[N][3];
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 3; c++) sp[b][c] = s.p[b][c], sv[b][c] = s.v[b][c];

#pragma unroll
        for (int k = 0; k < 6; k++) {
            acc_only<N, false>(sp, s.m, ka[k]);
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    kv[k][b][c] = sv[b][c];
                    double sumv, suma;
#define KV(i) kv[i][b][c]
#define KA(i) ka[i][b][c]
                    if (k == 0) {
                        sp[b][c] = s.p[b][c] + h * b1 * KV(0);
                        sv[b][c] = s.v[b][c] + h * b1 * KA(0);
                        continue;
                    } else if (k == 1) {
                        sumv = b2[0] * KV(0) + b2[1] * KV(1);
                        suma = b2[0] * KA(0) + b2[1] * KA(1);
                    } else if (k == 2) {
                        sumv = b3[0] * KV(0) + b3[1] * KV(1) + b3[2] * KV(2);
                        suma = b3[0] * KA(0) + b3[1] * KA(1) + b3[2] * KA(2);
                    } else if (k == 3) {
                        sumv = b4[0] * KV(0) + b4[1] * KV(1) + b4[2] * KV(2) + b4[3] * KV(3);
                        suma = b4[0] * KA(0) + b4[1] * KA(1) + b4[2] * KA(2) + b4[3] * KA(3);
                    } else if (k == 4) {
                        sumv = b5[0] * KV(0) + b5[1] * KV(1) + b5[2] * KV(2) + b5[3] * KV(3) + b5[4] * KV(4);
                        suma = b5[0] * KA(0) + b5[1] * KA(1) + b5[2] * KA(2) + b5[3] * KA(3) + b5[4] * KA(4);
                    } else {
                        sumv = b6[0] * KV(0) + b6[1] * KV(1) + b6[2] * KV(2) + b6[3] * KV(3) +
                               b6[4] * KV(4) + b6[5] * KV(5);
                        suma = b6[0] * KA(0) + b6[1] * KA(1) + b6[2] * KA(2) + b6[3] * KA(3) +
                               b6[4] * KA(4) + b6[5] * KA(5);
                    }
                    sp[b][c] = s.p[b][c] + h * sumv;
                    sv[b][c] = s.v[b][c] + h * suma;
                }
        }
        // sp, sv now hold p6, v6
        bool accept_step = true;
        if (ADAPTIVE) { // L204-269
            const float step_grow_power = -1. / (5 + 1.);
            const float step_shrink_power = -1. / 5;
            const float step_guess_safety_factor = 0.9;
            const float step_grow_max_factor = 5.0;
            const float step_shrink_min_factor = 0.2;
            double max_error = 0;
#pragma unroll
            for (int b = 0; b < N; b++) {
                double pem = 0, vem = 0, pm = 0, vm = 0;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double pe = h * (ecc[0] * KV(0) + ecc[1] * KV(1) + ecc[2] * KV(2) + ecc[3] * KV(3) +
                                           ecc[4] * KV(4) + ecc[5] * KV(5));
                    const double ve = h * (ecc[0] * KA(0) + ecc[1] * KA(1) + ecc[2] * KA(2) + ecc[3] * KA(3) +
                                           ecc[4] * KA(4) + ecc[5] * KA(5));
                    pem = (c == 0) ? pe * pe : pem + pe * pe;
                    vem = (c == 0) ? ve * ve : vem + ve * ve;
                    pm = (c == 0) ? sp[b][c] * sp[b][c] : pm + sp[b][c] * sp[b][c];
                    vm = (c == 0) ? sv[b][c] * sv[b][c] : vm + sv[b][c] * sv[b][c];
                }
                const double pen = pem / pm, ven = vem / vm;
                max_error = fmax(fmax(pen, ven), max_error);
            }
#undef KV
#undef KA
            const double normalized_error = max_error / ip.error_tolerance;
            const double step_guess_power = (normalized_error < 1.) ? step_grow_power : step_shrink_power;
            const double step_change_factor =
                ((normalized_error < 0.5) || (normalized_error > 1.0))
                    ? step_guess_safety_factor * pow(normalized_error, 0.5 * step_guess_power)
                    : 1.0;
            const double new_time_step =
                (normalized_error > 1.)
                    ? fmax(time_step * fmax(step_change_factor, (double)step_shrink_min_factor),
                           ip.min_time_step)
                    : fmin(time_step * fmax(fmin(step_change_factor, (double)step_grow_max_factor), 1.0),
                           ip.max_time_step);
            accept_step = (normalized_error < 1.0) || (fabs(time_step - new_time_step) < 1e-10);
            time_step = new_time_step;
        }
        attempted++;
        if (accept_step) {
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) s.p[b][c] = sp[b][c], s.v[b][c] = sv[b][c];
            s.t += h;
            accepted++;
        }
        // monitor runs only for lanes that accepted (L283); the call itself is wave-wide
        if (MON) mon_call(ms, s, mp, lg, accept_step);
        if (accept_step && s.state == SWARM_HIP_ACTIVE && s.t >= dest) s.state = SWARM_HIP_INACTIVE;
    }
    if (accepted || s.state != state_in) store_state(e, gid, s);
    if (cnt.accepted) cnt.accepted[gid] += accepted;
    if (cnt.attempted) cnt.attempted[gid] += attempted;
}

template <int N, bool ADAPTIVE>
static int launch_n(const LaunchCtx &c) {
    if (monitor_is_noop(c.mp))
        hipLaunchKernelGGL((rkck_kernel<N, ADAPTIVE, false>), dim3(c.grid), dim3(c.block), 0, c.stream, c.ens,
                           c.ip, c.mp, c.lg, c.cnt);
    else
        hipLaunchKernelGGL((rkck_kernel<N, ADAPTIVE, true>), dim3(c.grid), dim3(c.block), 0, c.stream, c.ens,
                           c.ip, c.mp, c.lg, c.cnt);
    return check_launch("rkck kernel launch");
}

int launch_rkck(const LaunchCtx &c, bool adaptive) {
    if (adaptive) {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, true>(c)));
    } else {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, false>(c)));
    }
    return SWARM_HIP_OK;
}

} // namespace swh
