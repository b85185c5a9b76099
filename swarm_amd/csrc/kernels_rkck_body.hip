// kernels_rkck_body.hip -- Cash-Karp Runge-Kutta (integrators/rkck_integrator.hpp:90-294, fixed and adaptive step) with
// ONE LANE PER BODY, for small ensembles.
//
// With one lane per system (kernels_rkck.hip) an attempt is one chain of six force evaluations over all pairs, and the
// six stage accelerations of all bodies (108 doubles at nbod 6) go through AGPRs; below one wavefront per SIMD the
// run time is that chain's latency whatever nsys is.  Here the nbod bodies of a system sit in nbod adjacent lanes
// (floor(64/nbod) systems per wavefront, one wavefront per workgroup): a lane keeps the 18 stage accelerations of its
// own body, evaluates the force on it from its partners' stage positions (rows in LDS, LaneExchange) and forms its own stage
// positions, solution and error estimate; the system's normalised error is the maximum over its lanes, and every lane
// of a system runs the same step control on it.  The reference's kernel is laid out the same way one level finer
// (3 nbod threads per system, L90-131).
//
// Arithmetic is that of the lane-per-system kernel operation for operation (rkck_common.hpp holds everything but the
// forces for both; forces accumulate the partners in increasing index order and the two signs of a pair in acc_only
// cancel), so both kernels give bit-identical ensembles, step sizes and accept/reject sequences
// (tests/test_gpu_rkck_body.py).  Serves the no-op monitor and stop_on_ejection without printing (deactivation, snapshot records).
#include "body_monitor.hpp"
#include "launch.hpp"
#include "rkck_common.hpp"

namespace swh {

template <int N, bool ADAPTIVE, int MONB>
__global__ __launch_bounds__(64) void rkck_body_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg, Counters cnt) {
    constexpr int G = W / N; // systems per wavefront
    constexpr bool EJ = MONB != MONB_OFF; // some monitor is bound
    constexpr bool ROLLED = BodyLoops<N>::ROLLED; // nbod 17 .. 32: partner loops rolled, masses in LDS (device_common.hpp)
    constexpr int BODY_TRIP = BodyLoops<N>::trip(ROLLED, 16);
    const int lane = threadIdx.x;
    const int g = lane / N, b = lane - g * N;
    int sys;
    const bool valid = body_slot_system(e, lane < G * N, (long long)blockIdx.x * G + g, sys);
    const int base = valid ? g * N : 0; // first lane of my system (idle lanes read system 0 of the wave)
    const unsigned long long group_mask = ((1ull << N) - 1) << base;

    const size_t chunk = sys / W, sl = sys % W;
    double *row = e.body + ((chunk * N + b) * e.nbf) * W + sl;
    double p[3], v[3], mo[ROLLED ? 1 : N - 1], t = 0.0;
    int state = SWARM_HIP_DISABLED;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        p[c] = valid ? row[c * W] : (c == 0 ? 1.0 + lane : 0.0); // idle lanes: distinct finite dummies
        v[c] = valid ? row[(3 + c) * W] : 0.0;
    }
    const double m_own = valid ? row[6 * W] : 1.0;
    const int id = (EJ && valid) ? e.sysi[chunk * 2 * W + W + sl] : 0;
    __shared__ double mass_row[W]; // ROLLED only: the mass of every lane's body
    if constexpr (ROLLED) {
        mass_row[lane] = m_own;
        __syncthreads();
    } else {
#pragma unroll
        for (int q = 0; q < N - 1; q++) {
            const int j = q < b ? q : q + 1; // partners in increasing index order
            mo[q] = valid ? e.body[((chunk * N + j) * e.nbf + 6) * W + sl] : 1.0;
        }
    }
    // mass of my q-th partner, who sits in lane `src`
    auto partner_mass = [&](const int q, const int src) {
        if constexpr (ROLLED)
            return mass_row[src];
        else
            return mo[q];
    };
    if (valid) {
        t = e.sysd[chunk * e.nsf * W + sl];
        state = e.sysi[chunk * 2 * W + sl];
    }
    const int state_in = state;

    // the force on my body with all bodies at `pos` (acc_only<N, false>); wave-wide, no divergence inside; the
    // positions travel through LDS (LaneExchange, device_common.hpp)
    typedef LaneExchange<3> Xchg;
    __shared__ __align__(16) typename Xchg::Row xchg[W];
    auto force = [&](const double (&pos)[3], double (&a)[3]) {
        Xchg::put(xchg, lane, pos);
        a[0] = a[1] = a[2] = 0.0;
        SWH_BODY_LOOP
        for (int q = 0; q < N - 1; q++) {
            double pj[3];
            const int src = base + (q < b ? q : q + 1);
            Xchg::get(xchg, src, pj);
            const double d0 = pj[0] - pos[0], d1 = pj[1] - pos[1], d2 = pj[2] - pos[2];
            const double r2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
            const double y = rsqrt_full(r2);
            const double rinv = y * (y * y);
            const double sc = rinv * partner_mass(q, src);
            a[0] = __builtin_fma(d0, sc, a[0]);
            a[1] = __builtin_fma(d1, sc, a[1]);
            a[2] = __builtin_fma(d2, sc, a[2]);
        }
        Xchg::done();
    };
    // the monitor of one system, called by the whole wavefront (body_monitor.hpp)
    BodyMonitor<N, MONB> monitor;
    monitor.init(t);
    auto ejection_monitor = [&](const bool take_part) {
        monitor(take_part, b, base, group_mask, p, v, m_own, [&](const int d) { return partner_mass(d, base + d); }, t,
                id, sys, state, mp, lg, [](bool, bool) {});
    };
    if (EJ) ejection_monitor(valid); // rkck_integrator.hpp:140: once before the loop, every system

    const int iteration_count = (int)(ip.destination_time / ip.max_time_step); // L75
    const double dest = ip.destination_time;
    double time_step = ip.max_time_step; // L133
    unsigned long long accepted = 0, attempted = 0;

    // Every lane goes through every attempt of the slowest system of its wave (the exchange needs all lanes); a system
    // that is done, or was not ACTIVE at launch, never commits.
    bool live = valid && state == SWARM_HIP_ACTIVE;
    for (int iter = 0; iter < iteration_count && __any(live); iter++) {
        double h = time_step;
        {
#pragma clang fp contract(off)
            if (t + h > dest) h = dest - t;
        }
        double ka[6][3], sp[3] = {p[0], p[1], p[2]};
        // stages 1..6: ka[k] = acc(p_k); p_{k+1} from (pos, vel, ka[0..k-1])
        static_for<6>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            force(sp, ka[k]);
            if constexpr (k < 5) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double kc[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) kc[j] = j < k ? ka[j][c] : 0.0;
                    sp[c] = rkck_stage_position<k>(h, p[c], v[c], kc);
                }
            }
        });
        // 5th-order solution and the embedded error estimate of my body (L193-199, L232-238)
        double p6[3], v6[3], pe[3], ve[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double kc[6];
#pragma unroll
            for (int j = 0; j < 6; j++) kc[j] = ka[j][c];
            rkck_solution<ADAPTIVE>(h, p[c], v[c], kc, p6[c], v6[c], pe[c], ve[c]);
        }
        bool accept_step = true;
        if (ADAPTIVE) {
            const double mine = rkck_body_error(pe, ve, p6, v6);
            double max_error = 0;
            SWH_BODY_LOOP
            for (int j = 0; j < N; j++) max_error = fmax(__shfl(mine, base + j), max_error);
            accept_step = rkck_step_control(max_error, time_step, ip); // L204-269, the same in every lane of the system
        }
        const bool stepped = accept_step && live;
        if (live) attempted++;
        if (stepped) {
#pragma unroll
            for (int c = 0; c < 3; c++) p[c] = p6[c], v[c] = v6[c];
            t += h;
            accepted++;
        }
        if (EJ) ejection_monitor(stepped); // the monitor runs for systems that accepted (L283); wave-wide ballot
        if (stepped && state == SWARM_HIP_ACTIVE && t >= dest) state = SWARM_HIP_INACTIVE;
        live = live && state == SWARM_HIP_ACTIVE;
    }
    if (valid && (accepted || state != state_in)) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            row[c * W] = p[c];
            row[(3 + c) * W] = v[c];
        }
        if (b == 0) {
            e.sysd[chunk * e.nsf * W + sl] = t;
            e.sysi[chunk * 2 * W + sl] = state;
        }
    }
    if (valid && b == 0) {
        if (cnt.accepted) cnt.accepted[sys] += accepted;
        if (cnt.attempted) cnt.attempted[sys] += attempted;
    }
}

// one wavefront per workgroup, so that a small ensemble spreads over as many CUs as it has wavefronts
template <int N>
static int launch_body_n(const LaunchCtx &c, bool adaptive) {
    constexpr int G = W / N;
    const int monb = body_monitor_of(c.mp, c.ens.nbod);
    BodyLaunch bl;
    if (int rc = prepare_body_launch(c, G, monb != MONB_OFF, bl)) return rc;
    // (the second launch: rkck's monitor call before the loop for the systems that are not ACTIVE, launch.hpp)
#define SWH_LAUNCH_BODY(AD, M)                                                                                         \
    do {                                                                                                               \
        hipLaunchKernelGGL((rkck_body_kernel<N, AD, M>), dim3(bl.grid), dim3(W), 0, c.stream, bl.view, c.ip, c.mp,     \
                           c.lg, c.cnt);                                                                               \
        if (bl.rest_grid)                                                                                              \
            hipLaunchKernelGGL((rkck_body_kernel<N, AD, M>), dim3(bl.rest_grid), dim3(W), 0, c.stream, bl.rest, c.ip,  \
                               c.mp, c.lg, c.cnt);                                                                     \
    } while (0)
#define SWH_LAUNCH_BODY_MON(AD)                                       \
    switch (monb) {                                                   \
    case MONB_EJECTION: SWH_LAUNCH_BODY(AD, MONB_EJECTION); break;    \
    case MONB_EJECTION_OR_CLOSE: SWH_LAUNCH_BODY(AD, MONB_EJECTION_OR_CLOSE); break; \
    case MONB_LOG_INTERVAL: SWH_LAUNCH_BODY(AD, MONB_LOG_INTERVAL); break; \
    default: SWH_LAUNCH_BODY(AD, MONB_OFF); break;                    \
    }
    if (adaptive) {
        SWH_LAUNCH_BODY_MON(true)
    } else {
        SWH_LAUNCH_BODY_MON(false)
    }
#undef SWH_LAUNCH_BODY_MON
#undef SWH_LAUNCH_BODY
    finish_body_launch(c, bl);
    return check_launch("rkck (lane per body) kernel launch");
}

int launch_rkck_body(const LaunchCtx &c, bool adaptive) {
    SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N>(c, adaptive)));
    return SWARM_HIP_OK;
}

} // namespace swh
