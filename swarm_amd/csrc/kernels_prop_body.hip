// kernels_prop_body.hip -- the tutorial propagators euler (third-order Taylor), midpoint (modified midpoint, 4 substeps),
// verlet (leapfrog with its own step) and hermite_prop (PEC2 Hermite as a propagator) of
// generic<Propagator,Monitor,Gravitation>::kernel
// (swarm/gpu/generic_gpu_bppt_integrator.hpp:137-238, propagators/{euler,midpoint,verlet,hermite_propagator}.hpp) with ONE
// LANE PER BODY, for small ensembles -- the layout of kernels_hermite_body.hip: floor(64/nbod) systems per wavefront,
// one wavefront per workgroup, a lane advances its own body and sums the forces of its partners, whose positions
// (and velocities) it reads from rows in LDS (LaneExchange).
//
// Arithmetic is that of the lane-per-system kernel (kernels_generic.hip) operation for operation: forces as
// accjerk<N> / acc_only<N, false> accumulate them per body, the per-coordinate updates from propagators_common.hpp
// and hermite_step<N>; both kernels give bit-identical ensembles (tests/test_gpu_prop_body.py).  Monitors:
// body_monitor.hpp.  Verlet's time step is one sum over all ordered pairs (i, j >= 1) of a system in a fixed order: a
// lane forms the terms of its own row, the rows go through LDS and every lane of the system adds all of them in that
// order.
#include "body_monitor.hpp"
#include "launch.hpp"
#include "propagators_common.hpp"

namespace swh {

template <int N, int KIND, int MONB, bool ROLLED>
__global__ __launch_bounds__(64) void prop_body_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg, Counters cnt) {
    constexpr int G = W / N; // systems per wavefront
    constexpr bool EJ = MONB != MONB_OFF;
    constexpr bool FOLD = N >= HERMITE_FOLD_FROM; // hermite_step<N>'s form from nbod 9 on
    // ROLLED: partner loops rolled, masses in LDS (device_common.hpp)
    constexpr int BODY_TRIP = BodyLoops<N>::trip(ROLLED, KIND == SWARM_HIP_VERLET ? 16 : 8);
    [[maybe_unused]] constexpr int BODY_TRIP_OUTER = BodyLoops<N>::outer_trip(ROLLED);
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

    typedef LaneExchange<6> Xchg6;
    typedef LaneExchange<3> Xchg3;
    __shared__ __align__(16) typename Xchg6::Row xchg6[W];
    __shared__ __align__(16) typename Xchg3::Row xchg3[W];
    // acceleration and jerk of my body from (pos, vel) of all lanes: accjerk<N>, first term by multiplication
    auto accjerk_body = [&](const double (&pos)[3], const double (&vel)[3], double (&a)[3], double (&jk)[3]) {
        {
            const double pv6[6] = {pos[0], pos[1], pos[2], vel[0], vel[1], vel[2]};
            Xchg6::put(xchg6, lane, pv6);
        }
        // (the unrolled form sets the sums with the first term's product; fma(x, y, -0.0) is that product bit for bit)
        if constexpr (ROLLED) a[0] = a[1] = a[2] = jk[0] = jk[1] = jk[2] = -0.0;
        SWH_BODY_LOOP
        for (int q = 0; q < N - 1; q++) {
            double o[6];
            const int src = base + (q < b ? q : q + 1);
            Xchg6::get(xchg6, src, o);
            const double d0 = o[0] - pos[0], d1 = o[1] - pos[1], d2 = o[2] - pos[2];
            const double e0 = o[3] - vel[0], e1 = o[4] - vel[1], e2 = o[5] - vel[2];
            const double r2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
            const double xv = __builtin_fma(d2, e2, __builtin_fma(d1, e1, d0 * e0));
            const double y = rsqrt_full(r2);
            const double y2 = y * y;
            const double rinv = y * y2;
            const double rv = (xv * 3.0) * y2;
            const double sc = rinv * partner_mass(q, src);
            const double t0 = __builtin_fma(-d0, rv, e0), t1 = __builtin_fma(-d1, rv, e1), t2 = __builtin_fma(-d2, rv, e2);
            if (!ROLLED && q == 0) {
                a[0] = d0 * sc, a[1] = d1 * sc, a[2] = d2 * sc;
                jk[0] = t0 * sc, jk[1] = t1 * sc, jk[2] = t2 * sc;
            } else {
                a[0] = __builtin_fma(d0, sc, a[0]), a[1] = __builtin_fma(d1, sc, a[1]), a[2] = __builtin_fma(d2, sc, a[2]);
                jk[0] = __builtin_fma(t0, sc, jk[0]), jk[1] = __builtin_fma(t1, sc, jk[1]);
                jk[2] = __builtin_fma(t2, sc, jk[2]);
            }
        }
        Xchg6::done();
    };
    // acceleration only: acc_only<N, false>, accumulated from zero
    auto acc_body = [&](const double (&pos)[3], double (&a)[3]) {
        Xchg3::put(xchg3, lane, pos);
        a[0] = a[1] = a[2] = 0.0;
        SWH_BODY_LOOP
        for (int q = 0; q < N - 1; q++) {
            double pj[3];
            const int src = base + (q < b ? q : q + 1);
            Xchg3::get(xchg3, src, pj);
            const double d0 = pj[0] - pos[0], d1 = pj[1] - pos[1], d2 = pj[2] - pos[2];
            const double r2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
            const double y = rsqrt_full(r2);
            const double rinv = y * (y * y);
            const double sc = rinv * partner_mass(q, src);
            a[0] = __builtin_fma(d0, sc, a[0]);
            a[1] = __builtin_fma(d1, sc, a[1]);
            a[2] = __builtin_fma(d2, sc, a[2]);
        }
        Xchg3::done();
    };
    // VerletPropagator::calc_timestep: my row of pair terms (partner planets only) into LDS, then the whole sum
    __shared__ double pair_terms[KIND == SWARM_HIP_VERLET ? N - 1 : 1][W];
    auto verlet_timestep_body = [&](const double (&pos)[3]) -> double {
        Xchg3::put(xchg3, lane, pos);
        SWH_BODY_LOOP
        for (int q = 0; q < N - 1; q++) {
            double pj[3];
            const int src = base + (q < b ? q : q + 1);
            Xchg3::get(xchg3, src, pj);
            // slot q is partner j = q (j < b) or q + 1; the star's lane has planets in every slot, a planet's slot 0
            // is the star, which is no term
            pair_terms[KIND == SWARM_HIP_VERLET ? q : 0][lane] =
                verlet_pair_term(pj[0] - pos[0], pj[1] - pos[1], pj[2] - pos[2], m_own + partner_mass(q, src));
        }
        __syncthreads(); // a single wavefront: the terms are stored before they are read
        double sum = 0.0;
        {
#pragma clang fp contract(off)
            SWH_BODY_LOOP_OUTER
            for (int i = 0; i < N; i++)
                SWH_BODY_LOOP
                for (int j = 1; j < N; j++) {
                    if (i == j) continue;
                    sum = sum + pair_terms[KIND == SWARM_HIP_VERLET ? (j < i ? j : j - 1) : 0][base + i];
                }
        }
        Xchg3::done();
        return verlet_step_of(sum, ip.timestep_scale, ip.max_timestep);
    };
    // the monitor of one system, called by the whole wavefront (body_monitor.hpp)
    BodyMonitor<N, MONB, ROLLED> monitor;
    monitor.init(t);
    auto call_monitor = [&](const bool take_part) {
        monitor(take_part, b, base, group_mask, p, v, m_own, [&](const int d) { return partner_mass(d, base + d); }, t,
                id, sys, state, mp, lg, [](bool, bool) {});
    };

    unsigned long long steps = 0;
    const bool mine = valid && state == SWARM_HIP_ACTIVE; // only these systems are integrated and stored
    if (__any(mine)) {
        const double dest = ip.destination_time;
        // HermitePropagator::init, hermite_propagator.hpp:77-82: (acc0, jerk0) carried across steps
        double a0[3], j0[3];
        StepConsts k;
        if (KIND == SWARM_HIP_HERMITE_PROP) {
            accjerk_body(p, v, a0, j0);
            k.set(ip.time_step);
        }
        double timestep = 0.0;
        if (KIND == SWARM_HIP_VERLET) timestep = verlet_timestep_body(p);
        bool live = mine;
        // Every lane walks through every step of the slowest system of its wave (the exchange needs all lanes); a
        // system that is done, or was not ACTIVE at launch, computes and discards.
        for (int iter = 0; iter < ip.max_iterations && __any(live); iter++) {
            double max_timestep;
            {
#pragma clang fp contract(off)
                max_timestep = dest - t;
            }
            double np[3], nv[3], h;
            if (KIND == SWARM_HIP_EULER) { // EulerPropagator::advance, euler.hpp:101-122
                h = fmin(ip.time_step, max_timestep);
                double a[3], jk[3];
                accjerk_body(p, v, a, jk);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    np[c] = p[c], nv[c] = v[c];
                    euler_advance(h, a[c], jk[c], np[c], nv[c]);
                }
            } else if (KIND == SWARM_HIP_HERMITE_PROP) { // HermitePropagator::advance, hermite_propagator.hpp:106-142
                h = fmin(ip.time_step, max_timestep);
                if (h != k.h) k.set(h);
                double pp[3], pv[3], dd0[3];
#pragma unroll
                for (int c = 0; c < 3; c++) { // hermite_step<N>, predictor
                    pp[c] = __builtin_fma(k.h, __builtin_fma(k.hh, __builtin_fma(k.h3, j0[c], a0[c]), v[c]), p[c]);
                    pv[c] = __builtin_fma(k.h, __builtin_fma(k.hh, j0[c], a0[c]), v[c]);
                    np[c] = pp[c];
                    nv[c] = pv[c];
                    pp[c] = __builtin_fma(k.kp_j0, j0[c], __builtin_fma(k.kp_a, a0[c], pp[c]));
                    dd0[c] = __builtin_fma(k.kv_a, a0[c], k.kv_j0 * j0[c]);
                    if (FOLD) pv[c] = pv[c] + dd0[c];
                }
                double a1[3], j1[3];
#pragma unroll
                for (int round = 0; round < 2; round++) {
                    accjerk_body(np, nv, a1, j1);
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        np[c] = __builtin_fma(k.kp_j1, j1[c], __builtin_fma(-k.kp_a, a1[c], pp[c]));
                        nv[c] = FOLD ? pv[c] + __builtin_fma(-k.kv_a, a1[c], k.kv_j1 * j1[c])
                                     : pv[c] + __builtin_fma(-k.kv_a, a1[c], __builtin_fma(k.kv_j1, j1[c], dd0[c]));
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; c++) a0[c] = live ? a1[c] : a0[c], j0[c] = live ? j1[c] : j0[c];
            } else if (KIND == SWARM_HIP_VERLET) { // VerletPropagator::advance, verlet.hpp:139-175 (the step is not clipped)
                const double h_first_half = 0.5 * timestep;
                double a[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    np[c] = p[c], nv[c] = v[c];
                    verlet_first(h_first_half, v[c], np[c]);
                }
                acc_body(np, a);
                const double next = verlet_timestep_body(np);
                const double h_second_half = 0.5 * next;
#pragma unroll
                for (int c = 0; c < 3; c++) verlet_second(h_first_half, h_second_half, a[c], np[c], nv[c]);
                {
#pragma clang fp contract(off)
                    h = h_first_half + h_second_half;
                }
                if (live) timestep = next;
            } else { // MidpointPropagator::advance, midpoint.hpp:106-160
                const double H = fmin(max_timestep, ip.time_step);
                const int n = 4;
                const double hs = H / n;
                h = H;
                double p_i[3], p_im1[3], p_im2[3], v_i[3], v_im1[3], v_im2[3], a[3];
#pragma unroll
                for (int c = 0; c < 3; c++) p_im1[c] = p[c], v_im1[c] = v[c];
                acc_body(p_im1, a);
#pragma unroll
                for (int c = 0; c < 3; c++) midpoint_first(hs, p_im1[c], v_im1[c], a[c], p_i[c], v_i[c]);
#pragma unroll
                for (int i = 2; i <= n; i++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        p_im2[c] = p_im1[c], p_im1[c] = p_i[c];
                        v_im2[c] = v_im1[c], v_im1[c] = v_i[c];
                    }
                    acc_body(p_im1, a);
#pragma unroll
                    for (int c = 0; c < 3; c++) midpoint_next(hs, p_im2[c], v_im2[c], v_im1[c], a[c], p_i[c], v_i[c]);
                }
                acc_body(p_i, a);
#pragma unroll
                for (int c = 0; c < 3; c++) midpoint_last(hs, p_i[c], p_im1[c], v_i[c], v_im1[c], a[c], np[c], nv[c]);
            }
            if (live) {
#pragma unroll
                for (int c = 0; c < 3; c++) p[c] = np[c], v[c] = nv[c];
                {
#pragma clang fp contract(off)
                    t = t + h;
                }
                steps++;
            }
            if (EJ) call_monitor(live); // generic_gpu_bppt_integrator.hpp:189-224
            if (live && state == SWARM_HIP_ACTIVE && t >= dest) state = SWARM_HIP_INACTIVE;
            live = live && state == SWARM_HIP_ACTIVE;
        }
        if (mine) {
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
    }
    if (valid && b == 0) {
        if (cnt.accepted) cnt.accepted[sys] += steps;
        if (cnt.attempted) cnt.attempted[sys] += steps;
    }
}

template <int N, int KIND, int M, bool R>
static void enqueue(const LaunchCtx &c, const BodyLaunch &bl) {
    hipLaunchKernelGGL((prop_body_kernel<N, KIND, M, R>), dim3(bl.grid), dim3(W), 0, c.stream, bl.view, c.ip, c.mp, c.lg,
                       c.cnt);
}
// the loop form: the only one built for this nbod, or the launcher's choice where both are (BodyLoops,
// device_common.hpp): euler, midpoint and hermite_prop at nbod 13 .. 16; verlet's rolled form is behind at every size
template <int N, int KIND>
constexpr bool both_forms = KIND != SWARM_HIP_VERLET && N >= 13 && N <= BODY_UNROLL_MAX;
template <int N, int KIND, int M>
static void enqueue_form(const LaunchCtx &c, const BodyLaunch &bl, const bool rolled) {
    if constexpr (both_forms<N, KIND>) {
        if (rolled)
            enqueue<N, KIND, M, true>(c, bl);
        else
            enqueue<N, KIND, M, false>(c, bl);
    } else {
        enqueue<N, KIND, M, BodyLoops<N>::ROLLED>(c, bl);
    }
}

// one wavefront per workgroup, so that a small ensemble spreads over as many CUs as it has wavefronts
template <int N, int KIND>
static int launch_body_n(const LaunchCtx &c) {
    constexpr int G = W / N;
    BodyLaunch bl;
    if (int rc = prepare_body_launch(c, G, false, bl)) return rc;
    // nbod 13 .. 16: the rolled form from this many ACTIVE systems on (profiles/r3/prop_rolled_sweep.txt)
    static const int from_12288[BODY_UNROLL_MAX + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 12288, 12288, 12288, 12288};
    static const int from_8192[BODY_UNROLL_MAX + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8192, 8192, 8192, 8192};
    const bool rolled = rolled_form(c, KIND == SWARM_HIP_HERMITE_PROP ? from_8192 : from_12288);
#define SWH_LAUNCH_BODY(M) enqueue_form<N, KIND, M>(c, bl, rolled)
    switch (body_monitor_of(c.mp, c.ens.nbod)) {
    case MONB_EJECTION: SWH_LAUNCH_BODY(MONB_EJECTION); break;
    case MONB_EJECTION_OR_CLOSE: SWH_LAUNCH_BODY(MONB_EJECTION_OR_CLOSE); break;
    case MONB_LOG_INTERVAL: SWH_LAUNCH_BODY(MONB_LOG_INTERVAL); break;
    default: SWH_LAUNCH_BODY(MONB_OFF); break;
    }
#undef SWH_LAUNCH_BODY
    finish_body_launch(c, bl);
    return check_launch("propagator (lane per body) kernel launch");
}

int launch_prop_body(const LaunchCtx &c, int integrator) {
    switch (integrator) {
    case SWARM_HIP_EULER: SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N, SWARM_HIP_EULER>(c))); break;
    case SWARM_HIP_MIDPOINT: SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N, SWARM_HIP_MIDPOINT>(c))); break;
    case SWARM_HIP_HERMITE_PROP: SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N, SWARM_HIP_HERMITE_PROP>(c))); break;
    case SWARM_HIP_VERLET: SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N, SWARM_HIP_VERLET>(c))); break;
    default: set_error("no lane-per-body kernel for propagator id %d", integrator); return SWARM_HIP_EINVAL;
    }
    return SWARM_HIP_OK;
}

} // namespace swh
