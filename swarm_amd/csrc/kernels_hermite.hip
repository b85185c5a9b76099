// kernels_hermite.hip -- fused PEC2 Hermite kernels (fixed and adaptive step) for gfx950.
//
// Replaces hermite<Monitor,Grav>::kernel (integrators/hermite_integrator.hpp:74-187) and
// hermite_adap<Monitor,Grav>::kernel (integrators/hermite_adap.hpp:68-201) of the reference:
// predictor, two force evaluations + correctors, time update, monitor and end-of-integration
// test are one loop body over register-resident state; K steps per launch.
#include "body_monitor.hpp"
#include "launch.hpp"

namespace swh {


// The monitor proper, out of line, for the fixed-step kernel with stop_on_ejection or log_time_interval (MON_ON): it is reached only when the wave-uniform early-out in the step fails, through a stack copy of the state
// so that `s` itself never has its address taken.  Inlined, its code costs the two-step loop registers.
template <int N>
__device__ __attribute__((noinline)) void hermite_mon_call(MonState *ms, State<N> *s, const MonParams *mp,
                                                           const LogBuf *lg) {
    mon_call<N, false>(*ms, *s, *mp, *lg, true);
}

template <int N, bool ADAPTIVE, int MON>
__global__ __launch_bounds__(256) void hermite_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
                                                      Counters cnt) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    State<N> s;
    load_state(e, gid, s);

    const int state_in = s.state;
    MonState ms;
    mon_init(ms, s);
    double radius[N]; // log_transit: body attribute 0
    const bool have_radius = e.nbf > 7;
    if (MON == MON_TRANSIT) {
#pragma unroll
        for (int b = 0; b < N; b++)
            radius[b] = have_radius ? e.body[((size_t)(gid / W) * N + b) * e.nbf * W + (size_t)7 * W + gid % W] : 0.0;
    }
    // hermite_adap.hpp:130: the adaptive kernel calls the monitor once before the loop,
    // for every system of the ensemble whatever its state.
    if (ADAPTIVE && MON == MON_TRANSIT) {
        double a0[N][3], j0[N][3];
        accjerk<N>(s.p, s.v, s.m, a0, j0);
        transit_monitor<N>(s, a0, j0, radius, have_radius, mp, lg);
    } else if (ADAPTIVE && MON)
        mon_call<N, MON == MON_VERBOSE>(ms, s, mp, lg, true);

    unsigned long long steps = 0;
    bool stored = false; // the state has been written back already (general loop)
    // WHOLE: the wave is kept whole through ragged ends (general loop below).  Not for the fixed-step kernel with
    // nothing to monitor -- Hermite.cfg as shipped, the headline: its lanes all take the same steps unless their start
    // times differ, and the extra control flow cost that kernel 3 % (16.88 -> 17.38 ms, A/B on one box).
    constexpr bool WHOLE = ADAPTIVE || MON != MON_OFF;
    // ... but a wave that would start with 8 or fewer lanes takes ghosts along (device_common.hpp)
    const bool ghost = !WHOLE && take_ghosts<N>(s.state == SWARM_HIP_ACTIVE, s);
    if (WHOLE ? __any(s.state == SWARM_HIP_ACTIVE) : (s.state == SWARM_HIP_ACTIVE)) {
        double fa[2][N][3], fj[2][N][3]; // ping-pong force buffers
        accjerk<N>(s.p, s.v, s.m, fa[0], fj[0]);
        const double dest = ip.destination_time;
        StepConsts k;
        k.set(ip.time_step);

        // everything of a step after h is known: step, time, monitor, end test
        // `on` = this lane really takes the step; a lane that only keeps the wave whole comes with h = 0
        auto finish_step = [&](const double h, const double (&a0)[N][3], const double (&j0)[N][3],
                               double (&a1)[N][3], double (&j1)[N][3], const bool on) {
            hermite_step<N>(s, k, a0, j0, a1, j1);
            if (!on) return;
            s.t += h;
            steps++;
            if (MON == MON_TRANSIT)
                transit_monitor<N>(s, a1, j1, radius, have_radius, mp, lg);
            else if (MON == MON_EJECT_QUIET) {
                // stop_on_ejection, deactivation only: early-out, then the bare test (no records, no printf, no call)
                bool maybe = false;
                {
#pragma clang fp contract(off)
#pragma unroll
                    for (int b = 1; b < N; b++) {
                        maybe = maybe || ej_maybe_body(s.p[b], s.v[b], mp);
                    }
                }
                if (__any(maybe)) {
                    bool met = false;
#pragma unroll
                    for (int b = 1; b < N; b++) met = met || ej_test_quiet(s.p[b], s.v[b], s.m[b], mp);
                    if (met && mp.ej_deactivate) s.state = SWARM_HIP_DISABLED;
                }
            } else if (MON == MON_ON && !ADAPTIVE) { // adaptive steps differ per lane: some lane's snapshot is due on most steps
                bool maybe = false;
                if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) {
#pragma clang fp contract(off)
                    maybe = mp.log_interval != 0. && s.t >= ms.next_log_time;
                } else if (mp.ej_any) { // stop_on_ejection; the composite with a close-encounter part runs as MON_VERBOSE
#pragma clang fp contract(off)
#pragma unroll
                    for (int b = 1; b < N; b++) {
                        maybe = maybe || ej_maybe_body(s.p[b], s.v[b], mp);
                    }
                }
                if (__any(maybe)) {
                    State<N> tmp = s;
                    MonState mtmp = ms;
                    hermite_mon_call<N>(&mtmp, &tmp, &mp, &lg);
                    s = tmp;
                    ms = mtmp;
                }
            } else if (MON)
                mon_call<N, MON == MON_VERBOSE>(ms, s, mp, lg, true);
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
                // a wave thinned out to 8 lanes or fewer (ejections, different start times) would run its FP64
                // instructions at a third of the rate: hand the stragglers to the general loop, which keeps the wave whole
                if (WHOLE && __popcll(__ballot(true)) <= 8) break;
                finish_step(dt, fa[0], fj[0], fa[1], fj[1], true);
                iter++;
                if (full()) {
                    finish_step(dt, fa[1], fj[1], fa[0], fj[0], true);
                    iter++;
                } else {
#pragma unroll
                    for (int b = 0; b < N; b++)
#pragma unroll
                        for (int c = 0; c < 3; c++) fa[0][b][c] = fa[1][b][c], fj[0][b][c] = fj[1][b][c];
                }
            }
        }
        // General loop: per-step h (adaptive step, or the clipped final step of the fixed-step form).
        // The wave stays whole until its last system is done: a lane with nothing left to do takes steps of h = 0,
        // which leave its registers exactly as they are (every corrector weight is 0, p + 0 = p), instead of leaving
        // the loop -- MI355X issues FP64 VALU instructions at less than a third of the rate once 8 or fewer lanes of a
        // wavefront are enabled (tools/ubench_exec.hip), so a thinned-out wave would crawl through its tail.
        auto live = [&]() { return iter < ip.max_iterations && s.state == SWARM_HIP_ACTIVE; };
        constexpr bool PINGPONG = ADAPTIVE || WHOLE;
        if constexpr (PINGPONG) {
            // one step of the general loop: forces at the start of the step in (a0, j0), at its end in (a1, j1)
            auto general_step = [&](const double (&a0)[N][3], const double (&j0)[N][3], double (&a1)[N][3], double (&j1)[N][3]) {
                const bool on = !WHOLE || live();
                // A system that is done hands its state back at once: from here on its lane only keeps the wave whole, and
                // whatever those h = 0 steps leave in its registers -- unchanged values as long as its forces are finite,
                // NaN otherwise (0 * inf) -- is never written back.
                if (WHOLE && !on && !stored) {
                    if (steps || s.state != state_in) store_state(e, gid, s);
                    stored = true;
                }
                double h;
                if (ADAPTIVE) {
                    // calc_adaptive_time_step, hermite_adap.hpp:69-98: sum over bodies of |jerk|^2/|acc|^2
                    double tf = 0.0;
#pragma unroll
                    for (int b = 0; b < N; b++) {
                        // (explicit fused form: what the compiler's contraction gives, fixed so that the lane-per-body
                        // kernel computes the same bits)
                        const double a2 = __builtin_fma(a0[b][2], a0[b][2], __builtin_fma(a0[b][1], a0[b][1], a0[b][0] * a0[b][0]));
                        const double jj = __builtin_fma(j0[b][2], j0[b][2], __builtin_fma(j0[b][1], j0[b][1], j0[b][0] * j0[b][0]));
                        const double ra = rcp_full(a2); // jj / a2 through the reciprocal, fused into the running sum
                        tf = (b == 0) ? jj * ra : __builtin_fma(jj, ra, tf);
                    }
                    h = __builtin_fma(rsqrt_full(tf), ip.time_step_factor, ip.min_time_step);
                } else {
                    h = ip.time_step;
                }
                {
#pragma clang fp contract(off)
                    if (s.t + h > dest) h = dest - s.t;
                }
                if (WHOLE && !on) h = 0.0;
                if (ADAPTIVE || !WHOLE || h != k.h) k.template set<ADAPTIVE>(h);
                finish_step(h, a0, j0, a1, j1, on);
                if (on) iter++;
            };
            // two steps per trip through the ping-pong force buffers, as in the fast path: copying the 6 N doubles of
            // (a1, j1) back into (a0, j0) after every step cost the adaptive kernel 3 % (nbod 3) to 18 % (nbod 10)
            while (WHOLE ? __any(live()) : live()) {
                general_step(fa[0], fj[0], fa[1], fj[1]);
                if (!(WHOLE ? __any(live()) : live())) break;
                general_step(fa[1], fj[1], fa[0], fj[0]);
            }
        } else {
            // The fixed-step kernel with nothing to monitor comes here for its clipped last step only; it keeps the
            // plain loop with the copy (written out, not through the lambda above: with it the headline's fast loop,
            // instruction for instruction the same, came out with another register assignment and 0.8 % slower).
            while (live()) {
                double h = ip.time_step;
                {
#pragma clang fp contract(off)
                    if (s.t + h > dest) h = dest - s.t;
                }
                k.set(h);
                finish_step(h, fa[0], fj[0], fa[1], fj[1], true);
                iter++;
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) fa[0][b][c] = fa[1][b][c], fj[0][b][c] = fj[1][b][c];
            }
        }
    }
    if (ghost) return;
    if (!stored && (steps || s.state != state_in)) store_state(e, gid, s);
    if (cnt.accepted) cnt.accepted[gid] += steps;
    if (cnt.attempted) cnt.attempted[gid] += steps;
}

template <int N, bool ADAPTIVE>
static int launch_n(const LaunchCtx &c) {
#define SWH_LAUNCH_VARIANT(V) \
    hipLaunchKernelGGL((hermite_kernel<N, ADAPTIVE, V>), dim3(c.grid), dim3(c.block), 0, c.stream, c.ens, c.ip, c.mp, c.lg, c.cnt)
    switch (monitor_variant(c.mp)) {
    case MON_OFF: SWH_LAUNCH_VARIANT(MON_OFF); break;
    case MON_ON:
        if (c.mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE && c.mp.ce_any)
            SWH_LAUNCH_VARIANT(MON_VERBOSE); // no cheap early-out: inline monitor (prints only if ce_verbose is set)
        else if (!ADAPTIVE && c.mp.kind == SWARM_HIP_MON_EJECTION && !c.mp.ej_log) {
            if constexpr (!ADAPTIVE) SWH_LAUNCH_VARIANT(MON_EJECT_QUIET);
        } else
            SWH_LAUNCH_VARIANT(MON_ON);
        break;
    case MON_TRANSIT:
        if constexpr (ADAPTIVE) { // plugins/hermite_adap.cu:61 is the only registration of log_transit
            SWH_LAUNCH_VARIANT(MON_TRANSIT);
        } else {
            set_error("monitor log_transit is bound to hermite_adap only (plugins/hermite_adap.cu:61)");
            return SWARM_HIP_EINVAL;
        }
        break;
    default: SWH_LAUNCH_VARIANT(MON_VERBOSE); break;
    }
#undef SWH_LAUNCH_VARIANT
    return check_launch("hermite kernel launch");
}

// One lane per body pays while the ensemble is too small to give every SIMD a wavefront of whole systems;
// beyond the measured crossover (tools/lane_sweep.sh, profiles/r2/hermite_lane_sweep.txt, DESIGN 5: nsys ~ 22k for
// nbod 3..7 and 9, 36k for nbod 8, 45k for nbod 10) the lane-per-system kernel's throughput wins.  Limits sit at the
// last grid point where the lane-per-body kernel still wins.
static bool small_ensemble(const LaunchCtx &c) {
    static const int limit[SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM + 1] = {0, 0, 0, 16384, 20480, 24576, 24576, 24576, 32768, 24576, 40960};
    return within_limit(c, limit);
}

int launch_hermite(const LaunchCtx &c, bool adaptive) {
    // small ensembles with nothing to monitor, with stop_on_ejection or the ejection-or-close-encounter composite
    // (deactivation and/or snapshot records, no printf) or with log_time_interval: one lane per body (kernels_hermite_body.hip), each lane testing its own planet and
    // writing its own body's part of a record
    const int layout = choose_layout(c, body_monitor_of(c.mp, c.ens.nbod) != MONB_NONE, small_ensemble(c));
    if (layout < 0) return layout;
    if (layout == 1) return launch_hermite_body(c, adaptive);
    if (adaptive) {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, true>(c)));
    } else {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, false>(c)));
    }
    return SWARM_HIP_OK;
}

} // namespace swh
