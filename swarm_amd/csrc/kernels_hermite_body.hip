// kernels_hermite_body.hip -- PEC2 Hermite (fixed and adaptive step) with ONE LANE PER BODY, for small ensembles.
//
// The one-lane-per-system kernel (kernels_hermite.hip) needs >= 65536 systems to put one wavefront
// on every SIMD; below that its run time is the latency of one wave's dependent steps whatever nsys
// is (the reference's saturation sweep, scripts/saturation: nsys = 32 ... 32768, is flat).  Here the
// nbod bodies of a system sit in nbod adjacent lanes (floor(64/nbod) systems per wavefront): a lane
// predicts and corrects its own body and sums the forces of its nbod-1 partners, whose predicted
// positions and velocities it reads from the partner lanes (rows in LDS, LaneExchange).  The dependent chain per
// step is about nbod/2.5 times shorter, at the price of computing every pair on both sides.
//
// Arithmetic is that of accjerk<N> / hermite_step<N> (device_common.hpp) operation for operation:
// a body accumulates its partners in increasing index order, first term by multiplication, and
// (p_j - p_i), (v_j - v_i) enter the pair quantities only through products that are even in the
// sign, so both kernels give bit-identical ensembles (tests/test_gpu_hermite.py).
// The loops over the partners exist in two forms (BodyLoops, device_common.hpp): unrolled with the masses in registers
// (ROLLED = false: nbod <= 16) and rolled with the masses in LDS (nbod >= 17, and nbod 12 .. 16 for large ensembles,
// where both are built and launch_body_n picks); the same operations in the same order.
// Used by launch_hermite for the no-op monitor (Hermite.cfg as shipped: no deactivate/log keys), MONB = 0; for
// stop_on_ejection without printing, MONB = 1: every lane tests its own planet (stop_on_ejection.hpp:108-138), the
// lanes of a system share the verdict through one ballot and, with log_on_ejection, write the snapshot record
// together (log_system_body); for log_time_interval, MONB = 2 (log_time_interval.hpp:98-119), every lane keeping
// its system's next_log_time; and for the ejection-or-close-encounter composite without printing, MONB = 3.
#include "body_monitor.hpp"
#include "launch.hpp"

namespace swh {

template <int N, int MONB, bool ADAPTIVE, bool ROLLED>
__global__ __launch_bounds__(64) void hermite_body_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg, Counters cnt) {
    constexpr bool EJ = MONB != MONB_OFF; // some monitor is bound
    constexpr int G = W / N; // systems per wavefront
    constexpr bool FOLD = N >= HERMITE_FOLD_FROM;
    constexpr int BODY_TRIP = BodyLoops<N>::trip(ROLLED); // ROLLED: partner loops rolled, masses in LDS (device_common.hpp)
    const int lane = threadIdx.x;
    const int g = lane / N, b = lane - g * N;
    int sys;
    const bool valid = body_slot_system(e, lane < G * N, (long long)blockIdx.x * G + g, sys);
    const int base = g * N; // first lane of my system

    // own body, the partners' masses, the system's time and state (every lane of a group holds a copy)
    const size_t chunk = sys / W, sl = sys % W;
    double *row = e.body + ((chunk * N + b) * e.nbf) * W + sl;
    double p[3], v[3], mo[ROLLED ? 1 : N - 1], t = 0.0;
    const double m_own = ((EJ || ROLLED) && valid) ? row[6 * W] : 1.0;
    const int id = (EJ && valid) ? e.sysi[(sys / W) * 2 * W + W + sys % W] : 0;
    const unsigned long long group_mask = ((1ull << N) - 1) << base; // lanes of my system (base + N <= 64 when valid)
    int state = SWARM_HIP_DISABLED;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        p[c] = valid ? row[c * W] : (c == 0 ? 1.0 + lane : 0.0); // idle lanes: distinct finite dummies
        v[c] = valid ? row[(3 + c) * W] : 0.0;
    }
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
    // The monitor of one system, called by the whole wavefront (body_monitor.hpp)
    BodyMonitor<N, MONB, ROLLED> monitor;
    monitor.init(t);
    auto ejection_monitor = [&](const bool take_part) {
        monitor(take_part, b, base, group_mask, p, v, m_own,
                [&](const int d) { return partner_mass(d, (base + d) & (W - 1)); }, t, id, sys, state, mp, lg,
                [](bool, bool) {});
    };
    // hermite_adap.hpp:130: the adaptive kernel calls the monitor once before the loop, whatever the state
    if (EJ && ADAPTIVE) ejection_monitor(valid);

    // forces on my body from the current p, v of all lanes; wave-wide, no divergence inside; positions and velocities
    // travel through LDS (LaneExchange, device_common.hpp)
    typedef LaneExchange<6> Xchg;
    __shared__ __align__(16) typename Xchg::Row xchg[W];
    auto forces = [&](double (&a)[3], double (&jk)[3]) {
        {
            const double pv6[6] = {p[0], p[1], p[2], v[0], v[1], v[2]};
            Xchg::put(xchg, lane, pv6);
        }
        // (the unrolled form sets the sums with the first term's product; fma(x, y, -0.0) is that product bit for bit)
        if constexpr (ROLLED) a[0] = a[1] = a[2] = jk[0] = jk[1] = jk[2] = -0.0;
        SWH_BODY_LOOP
        for (int q = 0; q < N - 1; q++) {
            double o[6];
            const int src = (base + (q < b ? q : q + 1)) & (W - 1);
            Xchg::get(xchg, src, o);
            const double d0 = o[0] - p[0], d1 = o[1] - p[1], d2 = o[2] - p[2];
            const double e0 = o[3] - v[0], e1 = o[4] - v[1], e2 = o[5] - v[2];
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
        Xchg::done();
    };

    unsigned long long steps = 0;
    double a0[3], j0[3];
    forces(a0, j0);
    const double dest = ip.destination_time;
    StepConsts k;
    k.set(ip.time_step);
    int iter = 0;
    bool active = valid && state == SWARM_HIP_ACTIVE && iter < ip.max_iterations;
    // Every lane walks through every step of the slowest system of its wave (the exchange needs all lanes);
    // lanes of finished or idle groups compute and discard.
    while (__any(active)) {
        double h = ip.time_step;
        if (ADAPTIVE) {
            // calc_adaptive_time_step, hermite_adap.hpp:69-98: |jerk|^2/|acc|^2 summed over the bodies in index order
            const double a2 = __builtin_fma(a0[2], a0[2], __builtin_fma(a0[1], a0[1], a0[0] * a0[0]));
            const double jj = __builtin_fma(j0[2], j0[2], __builtin_fma(j0[1], j0[1], j0[0] * j0[0]));
            const double ra = rcp_full(a2);
            double tf = __shfl(jj, base) * __shfl(ra, base);
            SWH_BODY_LOOP
            for (int j = 1; j < N; j++) tf = __builtin_fma(__shfl(jj, base + j), __shfl(ra, base + j), tf);
            h = __builtin_fma(rsqrt_full(tf), ip.time_step_factor, ip.min_time_step);
        }
        {
#pragma clang fp contract(off)
            if (t + h > dest) h = dest - t; // hermite_integrator.hpp:120-123
        }
        if (ADAPTIVE)
            k.template set<true>(h);
        else if (h != k.h)
            k.set(h);
        const double p_in[3] = {p[0], p[1], p[2]}, v_in[3] = {v[0], v[1], v[2]};
        double pp[3], pv[3], dd0[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { // hermite_step<N>, predictor
            pp[c] = __builtin_fma(k.h, __builtin_fma(k.hh, __builtin_fma(k.h3, j0[c], a0[c]), v[c]), p[c]);
            pv[c] = __builtin_fma(k.h, __builtin_fma(k.hh, j0[c], a0[c]), v[c]);
            p[c] = pp[c];
            v[c] = pv[c];
            pp[c] = __builtin_fma(k.kp_j0, j0[c], __builtin_fma(k.kp_a, a0[c], pp[c]));
            dd0[c] = __builtin_fma(k.kv_a, a0[c], k.kv_j0 * j0[c]);
            if (FOLD) pv[c] = pv[c] + dd0[c]; // hermite_step<N>'s form from nbod 9 on
        }
        double a1[3], j1[3];
#pragma unroll
        for (int round = 0; round < 2; round++) {
            forces(a1, j1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                p[c] = __builtin_fma(k.kp_j1, j1[c], __builtin_fma(-k.kp_a, a1[c], pp[c]));
                v[c] = FOLD ? pv[c] + __builtin_fma(-k.kv_a, a1[c], k.kv_j1 * j1[c])
                            : pv[c] + __builtin_fma(-k.kv_a, a1[c], __builtin_fma(k.kv_j1, j1[c], dd0[c]));
            }
        }
        auto end_test = [&]() {
            if (state == SWARM_HIP_ACTIVE) {
                // hermite_integrator.hpp:177-179 (`>` dest-1e-12), hermite_adap.hpp:191-193 (`>=`)
#pragma clang fp contract(off)
                if (ADAPTIVE ? (t >= dest) : (t > dest - 1e-12)) state = SWARM_HIP_INACTIVE;
            }
            active = state == SWARM_HIP_ACTIVE && iter < ip.max_iterations;
        };
        const bool stepped = active;
        if (active) {
#pragma unroll
            for (int c = 0; c < 3; c++) a0[c] = a1[c], j0[c] = j1[c];
            t += h;
            steps++;
            iter++;
            if (!EJ) end_test(); // one divergent region per step when there is nothing to monitor
        } else { // a finished system keeps its state while the rest of the wave goes on
#pragma unroll
            for (int c = 0; c < 3; c++) p[c] = p_in[c], v[c] = v_in[c];
        }
        if (EJ) { // stop_on_ejection::operator(): after the step, before the end-of-integration test; wave-wide ballot
            ejection_monitor(stepped);
            if (stepped) end_test();
        }
    }
    if (valid && (steps || state != state_in)) {
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
        if (cnt.accepted) cnt.accepted[sys] += steps;
        if (cnt.attempted) cnt.attempted[sys] += steps;
    }
}

template <int N, int M, bool AD, bool R>
static void enqueue(const LaunchCtx &c, const BodyLaunch &bl) {
    hipLaunchKernelGGL((hermite_body_kernel<N, M, AD, R>), dim3(bl.grid), dim3(W), 0, c.stream, bl.view, c.ip, c.mp, c.lg,
                       c.cnt);
    if (bl.rest_grid)
        hipLaunchKernelGGL((hermite_body_kernel<N, M, AD, R>), dim3(bl.rest_grid), dim3(W), 0, c.stream, bl.rest, c.ip,
                           c.mp, c.lg, c.cnt);
}
// the loop form: the only one built for this nbod, or the launcher's choice where both are (BodyLoops, device_common.hpp)
template <int N>
constexpr bool both_forms = N >= 12 && N <= BODY_UNROLL_MAX;
template <int N, int M, bool AD>
static void enqueue_form(const LaunchCtx &c, const BodyLaunch &bl, const bool rolled) {
    if constexpr (both_forms<N>) {
        if (rolled)
            enqueue<N, M, AD, true>(c, bl);
        else
            enqueue<N, M, AD, false>(c, bl);
    } else {
        enqueue<N, M, AD, BodyLoops<N>::ROLLED>(c, bl);
    }
}

// one wavefront per workgroup, so that a small ensemble spreads over as many CUs as it has wavefronts
template <int N>
static int launch_body_n(const LaunchCtx &c, bool adaptive) {
    constexpr int G = W / N;
    const int monb = body_monitor_of(c.mp, c.ens.nbod);
    BodyLaunch bl;
    if (int rc = prepare_body_launch(c, G, adaptive && monb != MONB_OFF, bl)) return rc;
    // (the second launch: hermite_adap's monitor call before the loop for the systems that are not ACTIVE, launch.hpp)
    // nbod 12 .. 16: the rolled form from this many ACTIVE systems on (profiles/r3/body_rolled_sweep_nbod_11_16.txt)
    static const int rolled_from[BODY_UNROLL_MAX + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 16384, 8192, 8192, 6144, 6144};
    const bool rolled = rolled_form(c, rolled_from);
#define SWH_LAUNCH_BODY(M, AD) enqueue_form<N, M, AD>(c, bl, rolled)
    if (adaptive) {
        if (monb == MONB_EJECTION)
            SWH_LAUNCH_BODY(MONB_EJECTION, true);
        else if (monb == MONB_EJECTION_OR_CLOSE)
            SWH_LAUNCH_BODY(MONB_EJECTION_OR_CLOSE, true);
        else if (monb == MONB_LOG_INTERVAL)
            SWH_LAUNCH_BODY(MONB_LOG_INTERVAL, true);
        else
            SWH_LAUNCH_BODY(MONB_OFF, true);
    } else {
        if (monb == MONB_EJECTION)
            SWH_LAUNCH_BODY(MONB_EJECTION, false);
        else if (monb == MONB_EJECTION_OR_CLOSE)
            SWH_LAUNCH_BODY(MONB_EJECTION_OR_CLOSE, false);
        else if (monb == MONB_LOG_INTERVAL)
            SWH_LAUNCH_BODY(MONB_LOG_INTERVAL, false);
        else
            SWH_LAUNCH_BODY(MONB_OFF, false);
    }
#undef SWH_LAUNCH_BODY
    finish_body_launch(c, bl);
    return check_launch("hermite (lane per body) kernel launch");
}

int launch_hermite_body(const LaunchCtx &c, bool adaptive) {
    SWH_DISPATCH_NBOD_BODY(c.ens.nbod, return (launch_body_n<N>(c, adaptive)));
    return SWARM_HIP_OK;
}

} // namespace swh
