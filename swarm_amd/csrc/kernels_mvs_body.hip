// kernels_mvs_body.hip -- the MVS propagator (propagators/mvs.hpp, keplerian.hpp) with ONE LANE PER BODY, for small
// ensembles.
//
// The one-lane-per-system kernel (kernels_generic.hip) needs >= 131072 systems to give every SIMD its two wavefronts;
// below that a launch lasts as long as one wave's chain of dependent steps whatever nsys is, and most of that chain
// is the Kepler solver of one planet after the other.  Here the nbod bodies of a system sit in nbod adjacent lanes
// (floor(64/nbod) systems per wavefront, one wavefront per workgroup): every planet's lane solves its own Kepler
// problem, all at once, and sums the accelerations of its nbod-2 fellow planets, whose positions it reads from the
// partner lanes (rows in LDS, LaneExchange); the star's lane carries the barycentre and otherwise rides along.  The sums over all
// bodies of a system (coordinate changes, the star's drift) are formed by every lane of the system, in index order.
//
// Arithmetic is that of the lane-per-system kernel operation for operation: mvs_common.hpp holds the solver for both,
// the sums run in the same order with the same fused operations, and a planet accumulates its partners in increasing
// index order (for a pair the two signs of acc_only<N, true> cancel: fma(-d, -s, a) == fma(d, s, a)).  Both kernels
// give bit-identical ensembles (tests/test_gpu_mvs_body.py).
// Used by launch_generic for the no-op monitor (MVS.cfg as shipped) and, EJ = true, for stop_on_ejection without
// printing: every lane tests its own planet in the propagator's coordinates (generic_gpu_bppt_integrator.hpp:189-224
// calls the monitor between init() and shutdown()); a system that logs is converted for the record and back.
#include "body_monitor.hpp"
#include "launch.hpp"
#include "mvs_common.hpp"

namespace swh {

template <int N, int MONB, bool ROLLED>
__global__ __launch_bounds__(64) void mvs_body_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg, Counters cnt) {
    constexpr int G = W / N; // systems per wavefront
    constexpr bool EJ = MONB != MONB_OFF; // some monitor is bound
    constexpr int BODY_TRIP = BodyLoops<N>::trip(ROLLED); // ROLLED: loops over the bodies rolled, masses in LDS
    const int lane = threadIdx.x;
    const int g = lane / N, b = lane - g * N;
    int sys;
    const bool valid = body_slot_system(e, lane < G * N, (long long)blockIdx.x * G + g, sys);
    const int base = valid ? g * N : 0;      // first lane of my system (idle lanes read system 0 of the wave)
    const unsigned long long group_mask = ((1ull << N) - 1) << base;

    const size_t chunk = sys / W, sl = sys % W;
    double *row = e.body + ((chunk * N + b) * e.nbf) * W + sl;
    double p[3], v[3], m[ROLLED ? 1 : N], t = 0.0;
    int state = SWARM_HIP_DISABLED;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        p[c] = valid ? row[c * W] : (c == 0 ? 1.0 + lane : 0.0); // idle lanes: distinct finite dummies
        v[c] = valid ? row[(3 + c) * W] : 0.0;
    }
    __shared__ double mass_row[W]; // ROLLED only: the mass of every lane's body
    double m_own;
    if constexpr (ROLLED) {
        m_own = valid ? row[6 * W] : 1.0;
        mass_row[lane] = m_own;
        __syncthreads();
    } else {
#pragma unroll
        for (int j = 0; j < N; j++) m[j] = valid ? e.body[((chunk * N + j) * e.nbf + 6) * W + sl] : 1.0;
    }
    // mass of body j of my system
    auto mass_of = [&](const int j) {
        if constexpr (ROLLED)
            return mass_row[base + j];
        else
            return m[j];
    };
    if (valid) {
        t = e.sysd[chunk * e.nsf * W + sl];
        state = e.sysi[chunk * 2 * W + sl];
    }
    const int id = (EJ && valid) ? e.sysi[chunk * 2 * W + W + sl] : 0;
    double mo[(!ROLLED && N - 2 > 0) ? N - 2 : 1]; // the fellow planets' masses, in increasing index order
    if constexpr (!ROLLED) {
        m_own = m[0];
#pragma unroll
        for (int j = 1; j < N; j++) m_own = b == j ? m[j] : m_own;
#pragma unroll
        for (int q = 0; q < N - 2; q++) {
            const int k = (q + 1 < b) ? q + 1 : q + 2;
            mo[q] = m[1];
#pragma unroll
            for (int j = 2; j < N; j++) mo[q] = k == j ? m[j] : mo[q];
        }
    }

    const bool mine = valid && state == SWARM_HIP_ACTIVE; // only these systems are converted, integrated and stored
    const bool planet = mine && b >= 1;                   // this lane carries a planet that is being integrated
    bool live = mine;
    unsigned long long steps = 0;

    // MVSPropagator::convert_std_to_internal_coord / convert_internal_to_std_coord, mvs.hpp:159-190, 195-230
    // (mvs_to_internal / mvs_to_std of kernels_generic.hip: same sums, same order)
    auto to_internal = [&](const bool who) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double pc0 = __shfl(p[c], base);
            double sump = 0., sumv = 0., mtot = 0.;
            SWH_BODY_LOOP
            for (int j = 0; j < N; ++j) {
                const double mj = mass_of(j);
                mtot += mj;
                sump = __builtin_fma(mj, __shfl(p[c], base + j), sump);
                sumv = __builtin_fma(mj, __shfl(v[c], base + j), sumv);
            }
            sumv /= mtot;
            const double cp = sump / mtot;
            if (who) {
                v[c] = b == 0 ? sumv : v[c] - sumv;
                p[c] = b == 0 ? cp : p[c] - pc0;
            }
        }
    };
    auto to_std = [&](const bool who) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double sump = 0., sumv = 0.;
            const double m0 = mass_of(0), pc0 = __shfl(p[c], base), vc0 = __shfl(v[c], base);
            double mtot = m0;
            SWH_BODY_LOOP
            for (int j = 1; j < N; ++j) {
                const double mj = mass_of(j);
                mtot += mj;
                sump = __builtin_fma(mj, __shfl(p[c], base + j), sump);
                sumv = __builtin_fma(mj, __shfl(v[c], base + j), sumv);
            }
            sump /= mtot;
            const double dv0 = sumv / m0, dp = pc0 - sump;
            if (who) {
                p[c] = b == 0 ? p[c] - sump : p[c] + dp;
                v[c] = b == 0 ? v[c] - dv0 : v[c] + vc0;
            }
        }
    };
    // MVSPropagator::drift_step, mvs.hpp:242-256 (mvs_drift); the planets' velocities travel through LDS
    typedef LaneExchange<3> Xchg;
    __shared__ __align__(16) typename Xchg::Row xchg[W];
    // (mvs_momentum / mvs_drift of kernels_generic.hip: with no monitor bound the sum formed for the drift that ends a
    // step serves the drift that begins the next -- the velocities are the same -- which here also saves an exchange)
    double mv[3] = {0, 0, 0};
    auto momentum = [&]() {
#pragma clang fp contract(off)
        Xchg::put(xchg, lane, v);
        mv[0] = mv[1] = mv[2] = 0.0;
        SWH_BODY_LOOP
        for (int j = 1; j < N; ++j) {
            double vj[3];
            Xchg::get(xchg, base + j, vj);
            const double mj = mass_of(j);
#pragma unroll
            for (int c = 0; c < 3; c++) mv[c] = __builtin_fma(mj, vj[c], mv[c]);
        }
        Xchg::done();
    };
    auto drift = [&](const double hby2, const double inv_m0) {
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double shift = (mv[c] * hby2) * inv_m0;
            p[c] = b == 0 ? __builtin_fma(hby2, v[c], p[c]) : p[c] + shift;
        }
    };
    // GravitationAcc::sum_acc_planets (gravitation_acc.hpp:96-139) for my planet; acc_only<N, true>; the positions
    // travel through LDS (LaneExchange, device_common.hpp)
    auto planet_forces = [&](double (&a)[3]) {
        Xchg::put(xchg, lane, p);
        a[0] = a[1] = a[2] = 0.0;
        SWH_BODY_LOOP
        for (int q = 0; q < N - 2; q++) {
            double pj[3];
            const int k = (q + 1 < b) ? q + 1 : q + 2; // fellow planets in increasing index order
            Xchg::get(xchg, base + k, pj);
            const double d0 = pj[0] - p[0], d1 = pj[1] - p[1], d2 = pj[2] - p[2];
            const double r2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
            const double y = rsqrt_full(r2);
            const double rinv = y * (y * y);
            double mk;
            if constexpr (ROLLED)
                mk = mass_of(k);
            else
                mk = mo[q];
            const double sc = rinv * mk;
            a[0] = __builtin_fma(d0, sc, a[0]);
            a[1] = __builtin_fma(d1, sc, a[1]);
            a[2] = __builtin_fma(d2, sc, a[2]);
        }
        Xchg::done();
        if (!planet) a[0] = a[1] = a[2] = 0.0; // the star is not kicked
    };
    // drift_kepler of my planet (the star's and idle lanes go through the same instructions and discard)
    auto kepler = [&](const KeplerConsts &k) {
        KeplerOrbit o;
        kepler_begin(o, p, v, k);
        o.live = planet;
        double y, S, C;
        auto stumpff = [&]() {
            y = kepler_y(o);
            if (__builtin_expect(__all(!planet || fabs(y) < 1e-4), 1)) {
                SC_series(y, S, C);
            } else {
                double sc[2];
                SC_prussing_outlined(planet ? y : 0.0, sc);
                S = sc[0], C = sc[1];
            }
        };
        for (int i = 0; i < 7; i++) {
            if (!__any(kepler_more(o, i))) break;
            stumpff();
            kepler_pass(o, k, y, S, C);
        }
        stumpff();
        double np[3] = {p[0], p[1], p[2]}, nv[3] = {v[0], v[1], v[2]};
        kepler_end(o, k, y, S, C, ip.kepler_compensated != 0, planet, np, nv);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            p[c] = planet ? np[c] : p[c];
            v[c] = planet ? nv[c] : v[c];
        }
    };
    // The monitor of one system, called by the whole wavefront (body_monitor.hpp), in the propagator's coordinates; a
    // system that is to be logged goes to standard coordinates for the record and back, as generic_mon_call does it
    // (generic_gpu_bppt_integrator.hpp:189-224: pass_one, conversion, pass_two, log, conversion back).
    BodyMonitor<N, MONB, ROLLED> monitor;
    monitor.init(t);
    auto ejection_monitor = [&](const bool take_part) {
        monitor(take_part, b, base, group_mask, p, v, m_own, [&](const int d) { return mass_of(d); }, t, id, sys, state,
                mp, lg, [&](const bool who, const bool std_coordinates) {
                    if (std_coordinates)
                        to_std(who);
                    else
                        to_internal(who);
                });
    };

    if (__any(mine)) {
        const double dest = ip.destination_time;
        const double inv_m0 = 1.0 / mass_of(0);
        KeplerConsts kc;
        kc.set_star(sqrt(mass_of(0)));
        double acc[3];
        // prop.init()
        to_internal(mine);
        planet_forces(acc);
        momentum();
        // Every lane walks through every step of the slowest system of its wave (the exchange needs all lanes); the
        // lanes of a finished or idle system take steps of h = 0, which leave their registers as they are.
        for (int iter = 0; iter < ip.max_iterations && __any(live); iter++) {
            double max_timestep;
            {
#pragma clang fp contract(off)
                max_timestep = dest - t;
            }
            // MVSPropagator::advance, mvs.hpp:260-293
            const double hby2 = live ? 0.5 * fmin(max_timestep, ip.time_step) : 0.0;
            if (EJ) momentum(); // a monitor may have taken the system through standard coordinates
            drift(hby2, inv_m0);
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = __builtin_fma(hby2, acc[c], v[c]);
            kc.set_dt(2.0 * hby2);
            kepler(kc);
            planet_forces(acc);
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = __builtin_fma(hby2, acc[c], v[c]);
            momentum();
            drift(hby2, inv_m0);
            t += 2.0 * hby2;
            if (live) steps++;
            if (EJ) ejection_monitor(live); // generic_gpu_bppt_integrator.hpp:189-224, in the propagator's coordinates
            if (live && state == SWARM_HIP_ACTIVE && t >= dest) state = SWARM_HIP_INACTIVE;
            live = live && state == SWARM_HIP_ACTIVE;
        }
        // prop.shutdown()
        to_std(mine);
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

template <int N, int M, bool R>
static void enqueue(const LaunchCtx &c, const BodyLaunch &bl) {
    hipLaunchKernelGGL((mvs_body_kernel<N, M, R>), dim3(bl.grid), dim3(W), 0, c.stream, bl.view, c.ip, c.mp, c.lg, c.cnt);
}
// the loop form: the only one built for this nbod, or the launcher's choice where both are (BodyLoops, device_common.hpp)
template <int N>
constexpr bool both_forms = N >= 14 && N <= BODY_UNROLL_MAX;
template <int N, int M>
static void enqueue_form(const LaunchCtx &c, const BodyLaunch &bl, const bool rolled) {
    if constexpr (both_forms<N>) {
        if (rolled)
            enqueue<N, M, true>(c, bl);
        else
            enqueue<N, M, false>(c, bl);
    } else {
        enqueue<N, M, BodyLoops<N>::ROLLED>(c, bl);
    }
}

// one wavefront per workgroup, so that a small ensemble spreads over as many CUs as it has wavefronts
template <int N>
static int launch_body_n(const LaunchCtx &c) {
    constexpr int G = W / N;
    BodyLaunch bl;
    if (int rc = prepare_body_launch(c, G, false, bl)) return rc;
    // nbod 14 .. 16: the rolled form from this many ACTIVE systems on (profiles/r3/body_rolled_sweep_nbod_11_16.txt; at
    // 12 and 13 it is 2 - 8 % behind at every size)
    static const int rolled_from[BODY_UNROLL_MAX + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6144, 6144, 6144};
    const bool rolled = rolled_form(c, rolled_from);
#define SWH_LAUNCH_BODY(M) enqueue_form<N, M>(c, bl, rolled)
    switch (body_monitor_of(c.mp, c.ens.nbod)) {
    case MONB_EJECTION: SWH_LAUNCH_BODY(MONB_EJECTION); break;
    case MONB_EJECTION_OR_CLOSE: SWH_LAUNCH_BODY(MONB_EJECTION_OR_CLOSE); break;
    case MONB_LOG_INTERVAL: SWH_LAUNCH_BODY(MONB_LOG_INTERVAL); break;
    default: SWH_LAUNCH_BODY(MONB_OFF); break;
    }
#undef SWH_LAUNCH_BODY
    finish_body_launch(c, bl);
    return check_launch("mvs (lane per body) kernel launch");
}

int launch_mvs_body(const LaunchCtx &c) {
    SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N>(c)));
    return SWARM_HIP_OK;
}

} // namespace swh
