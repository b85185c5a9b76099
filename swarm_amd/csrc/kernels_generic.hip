// kernels_generic.hip -- propagator kernels: MVS, Euler(3rd-order Taylor), Verlet, Midpoint,
// Hermite-as-propagator.
//
// Replaces generic<Propagator,Monitor,Gravitation>::kernel
// (swarm/gpu/generic_gpu_bppt_integrator.hpp:137-238) and the propagators
// propagators/{mvs,keplerian,euler,verlet,midpoint}.hpp.  One lane per system, state in VGPRs.
//
// Deviations from the reference, on purpose:
//  * the reference converts a whole CUDA block to standard coordinates when ANY system of the
//    block needs to log (syncthreads_or, L192); here only the system that asked is converted;
//  * systems that are not ACTIVE at launch are left untouched (the reference runs MVS
//    init()/shutdown() coordinate round trips on them, which perturbs their last bits).
#include "body_monitor.hpp"
#include "launch.hpp"
#include "mvs_common.hpp"
#include "propagators_common.hpp"

namespace swh {

// `kepler_compensated=1` is a second MVS kernel, not a branch in the first: kepler_end's two forms of the f, g update
// differ in ~10 operations per planet, and told apart at run time (a wave-uniform branch per planet and coordinate)
// they cost the step 77 scalar branches and the moves that merge the two paths.  Internal id, not in the C-ABI.
constexpr int KIND_MVS_COMPENSATED = 100 + SWARM_HIP_MVS;
template <int KIND>
constexpr bool is_mvs = KIND == SWARM_HIP_MVS || KIND == KIND_MVS_COMPENSATED;

// Kepler drift of G planets in lockstep (the per-planet pieces are in mvs_common.hpp), with the Laguerre iterations of
// the G planets advancing together so that their G independent dependency chains interleave (one lane owns the whole
// system, so without this every iteration is one long serial chain of FP64 latencies).  The Stumpff functions take the
// series branch of SC_prussing for all planets with one wave-uniform test when every lane is in that regime
// (|alpha x^2| < 1e-4, the usual case for dt << orbital period), and the general per-planet routine otherwise.
template <int G>
__device__ __forceinline__ void SC_group(const double (&y)[G], double (&S)[G], double (&C)[G]) {
    bool small = true;
#pragma unroll
    for (int g = 0; g < G; g++) small = small && (fabs(y[g]) < 1e-4);
    if (__builtin_expect(__all(small), 1)) {
#pragma unroll
        for (int g = 0; g < G; g++) SC_series(y[g], S[g], C[g]);
    } else {
#pragma unroll
        for (int g = 0; g < G; g++) {
            double sc[2];
            SC_prussing_outlined(y[g], sc);
            S[g] = sc[0], C[g] = sc[1];
        }
    }
}

template <int N, int B0, int G, bool COMPENSATED>
__device__ __forceinline__ void drift_kepler_group(State<N> &s, const KeplerConsts &k) {
    KeplerOrbit o[G];
#pragma unroll
    for (int g = 0; g < G; g++) kepler_begin(o[g], s.p[B0 + g], s.v[B0 + g], k);
    // keplerian.hpp:143: the first three passes are unconditional ...
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double y[G], Sp[G], Cp[G];
#pragma unroll
        for (int g = 0; g < G; g++) y[g] = kepler_y(o[g]);
        SC_group<G>(y, Sp, Cp);
#pragma unroll
        for (int g = 0; g < G; g++) kepler_pass(o[g], k, y[g], Sp[g], Cp[g]);
    }
    // ... and up to four more follow, until x has stopped moving for every planet of the wavefront
    // (unrolled: a rolled tail costs 2 %, the whole loop rolled 17 %, C4 shard)
#pragma unroll
    for (int i = 3; i < 7; i++) {
        bool any = false;
#pragma unroll
        for (int g = 0; g < G; g++) any = kepler_more(o[g], i) || any;
        if (!__any(any)) break;
        double y[G], Sp[G], Cp[G];
#pragma unroll
        for (int g = 0; g < G; g++) y[g] = kepler_y(o[g]);
        SC_group<G>(y, Sp, Cp);
#pragma unroll
        for (int g = 0; g < G; g++) kepler_pass(o[g], k, y[g], Sp[g], Cp[g]);
    }
    double y[G], Sp[G], Cp[G];
#pragma unroll
    for (int g = 0; g < G; g++) y[g] = kepler_y(o[g]);
    SC_group<G>(y, Sp, Cp);
#pragma unroll
    for (int g = 0; g < G; g++) kepler_end(o[g], k, y[g], Sp[g], Cp[g], COMPENSATED, true, s.p[B0 + g], s.v[B0 + g]);
}

// all planets 1..N-1 in lockstep groups.  Pairs, an odd planet out joining the first group (3, 2, 2 ... instead of going
// alone at the end, where its passes would be one serial chain); four planets of nbod 5 and the first four of nbod 8's
// seven go together (4 and 4, 3).  Measured on MI355X, 131072 systems, ms per 1001 steps (round 3,
// profiles/r3/mvs_grouping_sweep.txt), pairs / balanced groups of at most 3 / of at most 4:
//   nbod 5  4.87 / 4.94 / 4.77     nbod 8  9.05 / 9.06 / 8.86     nbod 9  10.9 / 11.5 / 12.3     nbod 10  13.4 / 14.9 / 14.9
// (larger groups spill: 6 doubles per planet of a group are live across its passes); nbod 3, 4, 6, 7 within the noise.
constexpr int KEPLER_GROUP = 2;
constexpr int kepler_group_size(const int n, const int b0, const bool wide) {
    const int rem = n - b0;
    if (wide && b0 == 1 && (n == 5 || n == 8)) return 4;
    return (rem % KEPLER_GROUP == 1 && rem > 1) ? KEPLER_GROUP + 1 : (rem < KEPLER_GROUP ? rem : KEPLER_GROUP);
}
// WIDE: the groups of four; not with a monitor bound, whose kernel has no registers to spare (nbod 8 with the ejection
// monitor armed: 1440 instead of 800 bytes of scratch per lane, 18.6 instead of 15.9 ms on the C4 shard)
template <int N, bool COMPENSATED, bool WIDE, int B0 = 1>
__device__ __forceinline__ void drift_kepler_all(State<N> &s, const KeplerConsts &k) {
    if constexpr (B0 < N) {
        constexpr int G = kepler_group_size(N, B0, WIDE);
        drift_kepler_group<N, B0, G, COMPENSATED>(s, k);
        drift_kepler_all<N, COMPENSATED, WIDE, B0 + G>(s, k);
    }
}

// ---- MVS coordinate changes: propagators/mvs.hpp:159-190, 195-230 ----------------------------
template <int N>
__device__ __forceinline__ void mvs_to_internal(State<N> &s) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double pc0 = s.p[0][c];
        double sump = 0., sumv = 0., mtot = 0.;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double mj = s.m[j];
            mtot += mj;
            sump = __builtin_fma(mj, s.p[j][c], sump);
            sumv = __builtin_fma(mj, s.v[j][c], sumv);
        }
        sumv /= mtot;
        s.v[0][c] = sumv;
        s.p[0][c] = sump / mtot;
#pragma unroll
        for (int b = 1; b < N; b++) {
            s.v[b][c] -= sumv;
            s.p[b][c] -= pc0;
        }
    }
}
template <int N>
__device__ __forceinline__ void mvs_to_std(State<N> &s) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double sump = 0., sumv = 0.;
        const double m0 = s.m[0], pc0 = s.p[0][c], vc0 = s.v[0][c];
        double mtot = m0;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            const double mj = s.m[j];
            mtot += mj;
            sump = __builtin_fma(mj, s.p[j][c], sump);
            sumv = __builtin_fma(mj, s.v[j][c], sumv);
        }
        sump /= mtot;
        s.p[0][c] -= sump;
        s.v[0][c] -= sumv / m0;
#pragma unroll
        for (int b = 1; b < N; b++) {
            s.p[b][c] += pc0 - sump;
            s.v[b][c] += vc0;
        }
    }
}
// MVSPropagator::drift_step, mvs.hpp:242-256
// (`/ m0` is a multiplication by the launch-constant 1/m0: 1 ulp from the reference's quotient, 6 IEEE
// divides per step less)
// The planets' momentum sum of a drift is a function of the velocities alone, and nothing touches them between the
// drift that ends one step and the drift that begins the next: with no monitor bound the second sum of a step is kept
// for the first of the next (the same operations on the same operands, 3 (N-1) FMAs per step fewer).
template <int N>
__device__ __forceinline__ void mvs_momentum(const State<N> &s, double (&mv)[3]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        mv[c] = 0;
#pragma unroll
        for (int j = 1; j < N; ++j) mv[c] = __builtin_fma(s.m[j], s.v[j][c], mv[c]);
    }
}
template <int N>
__device__ __forceinline__ void mvs_drift(State<N> &s, const double (&mv)[3], double hby2, double inv_m0) {
#pragma clang fp contract(off)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        s.p[0][c] = __builtin_fma(hby2, s.v[0][c], s.p[0][c]);
        const double shift = (mv[c] * hby2) * inv_m0; // one product for all planets, then plain additions
#pragma unroll
        for (int b = 1; b < N; b++) s.p[b][c] = s.p[b][c] + shift;
    }
}

// Verlet time step: VerletPropagator::calc_timestep (verlet.hpp:112-137) over
// GravitationAcc::one_over_r (gravitation_acc.hpp:78-93), which returns |dx r^-3|^2 = r^-4.
template <int N>
__device__ __forceinline__ double verlet_timestep(const State<N> &s, const IntParams &ip) {
#pragma clang fp contract(off)
    double factor = 0.;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 1; j < N; ++j) {
            if (i == j) continue;
            factor = factor + verlet_pair_term(s.p[j][0] - s.p[i][0], s.p[j][1] - s.p[i][1], s.p[j][2] - s.p[i][2],
                                               s.m[i] + s.m[j]);
        }
    }
    return verlet_step_of(factor, ip.timestep_scale, ip.max_timestep);
}

// ---------------------------------------------------------------------------------------------
// The monitor of one step, out of line (generic_gpu_bppt_integrator.hpp:189-224): pass_one in internal coordinates,
// standard coordinates only for a system that is about to be logged.  Reached through a stack copy of the state.
template <int N, int KIND>
__device__ __attribute__((noinline)) void generic_mon_call(MonState *ms, State<N> *sp, const MonParams *mpp,
                                                           const LogBuf *lgp) {
    State<N> &s = *sp;
    const MonParams &mp = *mpp;
    const bool needs_std = mon_pass_one<N, false>(*ms, s, mp, *lgp);
    if (is_mvs<KIND> && needs_std) mvs_to_std(s);
    mon_pass_two(*ms, s, mp);
    const bool want = mon_need_log(*ms, mp);
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL || mp.ej_log || mp.ce_log)
        if (__any(want)) log_system(*lgp, s, want);
    if (is_mvs<KIND> && needs_std) mvs_to_internal(s);
}

// MVS keeps two waves per SIMD (<= 256 registers, a few spilled doubles from nbod 5 on): its Kepler solver is a
// chain of dependent FP64 operations with quarter-rate reciprocals, and a second resident wave hides them
// (131072 x 5..8: -15..20 %; at one wave of work per SIMD, 65536 systems, +2..3 %).
template <int N, int KIND, int MON>
__global__ __launch_bounds__(256, is_mvs<KIND> ? 2 : 1) void generic_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
                                                      Counters cnt) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    State<N> s;
    load_state(e, gid, s);
    unsigned long long steps = 0;
    const double dest = ip.destination_time;

    // The wave goes through the step loop whole, until its last system is done: a lane with nothing (left) to do takes
    // steps of h = 0, which leave its registers exactly as they are, instead of leaving the loop.  MI355X issues FP64
    // VALU instructions at less than a third of the rate once 8 or fewer lanes of a wavefront are enabled
    // (tools/ubench_exec.hip), so a wave thinned out by ejections or by different start times would crawl.
    // Done (WHOLE) where waves do thin out: with a monitor armed (ejections, close encounters) and for Verlet's
    // per-system step; with nothing to monitor every lane of a fixed-step propagator takes the same steps unless
    // start times differ, and the selects cost the uniform case 3 % (C4 shard 10.76 -> 11.11 ms).
    constexpr bool WHOLE = MON != MON_OFF || KIND == SWARM_HIP_VERLET;
    const bool mine = s.state == SWARM_HIP_ACTIVE; // only these systems are converted, integrated and stored
    // ... and where that leaves a wave 8 lanes or fewer, ghosts ride along (device_common.hpp)
    const bool ghost = !WHOLE && take_ghosts<N>(mine, s);
    if (WHOLE ? __any(mine) : (mine || ghost)) {
        bool live = mine || ghost;
        MonState ms;
        mon_init(ms, s);
        double acc[N][3];
        double timestep = 0.;
        const double inv_m0 = 1.0 / s.m[0];
        KeplerConsts kc;
        double mv[3] = {0., 0., 0.}; // MVS: the planets' momentum sum (mvs_momentum)
        // prop.init()
        if (is_mvs<KIND>) {
            kc.set_star(sqrt(s.m[0]));
            mvs_to_internal(s);
            acc_only<N, true>(s.p, s.m, acc);
            mvs_momentum(s, mv);
        } else if (KIND == SWARM_HIP_VERLET) {
            timestep = verlet_timestep(s, ip);
        }
        // HermitePropagator::init, hermite_propagator.hpp:77-82: (acc0, jerk0) carried across steps
        double hj[N][3], ha1[N][3], hj1[N][3];
        StepConsts hk;
        if (KIND == SWARM_HIP_HERMITE_PROP) {
            accjerk<N>(s.p, s.v, s.m, acc, hj);
            hk.set(ip.time_step);
        }

        bool stored = false;
        for (int iter = 0; iter < ip.max_iterations && (WHOLE ? __any(live) : live); iter++) {
            // A system that is done hands its state back at once (prop.shutdown() + store): from here on its lane only
            // keeps the wave whole, and whatever those h = 0 steps leave in its registers -- unchanged values as long
            // as its forces are finite, NaN otherwise (0 * inf) -- is never written back.
            if (WHOLE && mine && !live && !stored) {
                if (is_mvs<KIND>) mvs_to_std(s);
                store_state(e, gid, s);
                if (is_mvs<KIND>) mvs_to_internal(s);
                stored = true;
            }
            double max_timestep;
            {
#pragma clang fp contract(off)
                max_timestep = dest - s.t;
            }
            if (is_mvs<KIND>) { // MVSPropagator::advance, mvs.hpp:260-293
                const double hby2 = (!WHOLE || live) ? 0.5 * fmin(max_timestep, ip.time_step) : 0.0;
                if (MON != MON_OFF) mvs_momentum(s, mv); // a monitor may have taken the state through standard coordinates
                mvs_drift(s, mv, hby2, inv_m0);
#pragma unroll
                for (int b = 1; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) s.v[b][c] = __builtin_fma(hby2, acc[b][c], s.v[b][c]);
                if (2.0 * hby2 != kc.dt) kc.set_dt(2.0 * hby2);
                drift_kepler_all<N, KIND == KIND_MVS_COMPENSATED, MON == MON_OFF>(s, kc);
                acc_only<N, true>(s.p, s.m, acc);
#pragma unroll
                for (int b = 1; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) s.v[b][c] = __builtin_fma(hby2, acc[b][c], s.v[b][c]);
                mvs_momentum(s, mv);
                mvs_drift(s, mv, hby2, inv_m0);
                s.t += 2.0 * hby2;
            } else if (KIND == SWARM_HIP_HERMITE_PROP) { // HermitePropagator::advance, hermite_propagator.hpp:106-142
                const double h = (!WHOLE || live) ? fmin(ip.time_step, max_timestep) : 0.0;
                if (h != hk.h) hk.set(h);
                hermite_step<N>(s, hk, acc, hj, ha1, hj1);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) acc[b][c] = ha1[b][c], hj[b][c] = hj1[b][c];
                s.t += h;
            } else if (KIND == SWARM_HIP_EULER) { // EulerPropagator::advance, euler.hpp:101-122
                const double h = (!WHOLE || live) ? fmin(ip.time_step, max_timestep) : 0.0;
                double jerk[N][3];
                accjerk<N>(s.p, s.v, s.m, acc, jerk);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) euler_advance(h, acc[b][c], jerk[b][c], s.p[b][c], s.v[b][c]);
                s.t += h;
            } else if (KIND == SWARM_HIP_VERLET) { // VerletPropagator::advance, verlet.hpp:139-175
                const double h = (!WHOLE || live) ? timestep : 0.0;
                const double h_first_half = 0.5 * h;
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) verlet_first(h_first_half, s.v[b][c], s.p[b][c]);
                acc_only<N, false>(s.p, s.m, acc);
                timestep = verlet_timestep(s, ip);
                const double h_second_half = (!WHOLE || live) ? 0.5 * timestep : 0.0;
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) verlet_second(h_first_half, h_second_half, acc[b][c], s.p[b][c], s.v[b][c]);
                {
#pragma clang fp contract(off)
                    s.t = s.t + (h_first_half + h_second_half);
                }
            } else { // MidpointPropagator::advance, midpoint.hpp:106-160
                const double H = (!WHOLE || live) ? fmin(max_timestep, ip.time_step) : 0.0;
                const int n = 4;
                const double h = H / n;
                double p_i[N][3], p_im1[N][3], p_im2[N][3], v_i[N][3], v_im1[N][3], v_im2[N][3];
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) p_im1[b][c] = s.p[b][c], v_im1[b][c] = s.v[b][c];
                acc_only<N, false>(p_im1, s.m, acc);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        midpoint_first(h, p_im1[b][c], v_im1[b][c], acc[b][c], p_i[b][c], v_i[b][c]);
                    }
#pragma unroll
                for (int i = 2; i <= n; i++) {
#pragma unroll
                    for (int b = 0; b < N; b++)
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            p_im2[b][c] = p_im1[b][c], p_im1[b][c] = p_i[b][c];
                            v_im2[b][c] = v_im1[b][c], v_im1[b][c] = v_i[b][c];
                        }
                    acc_only<N, false>(p_im1, s.m, acc);
#pragma unroll
                    for (int b = 0; b < N; b++)
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            midpoint_next(h, p_im2[b][c], v_im2[b][c], v_im1[b][c], acc[b][c], p_i[b][c], v_i[b][c]);
                        }
                }
                acc_only<N, false>(p_i, s.m, acc);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        midpoint_last(h, p_i[b][c], p_im1[b][c], v_i[b][c], v_im1[b][c], acc[b][c], s.p[b][c], s.v[b][c]);
                    }
                s.t += H;
            }
            if (!WHOLE || live) steps++;

            // monitor (generic_gpu_bppt_integrator.hpp:189-224): pass_one in internal coordinates,
            // standard coordinates only for a system that is about to be logged.
            if (WHOLE && !live) {
                // nothing to monitor for a lane that stands still
            } else if (MON == MON_ON) {
                // stop_on_ejection / log_time_interval: wave-uniform early-out here, the monitor proper out of line
                // (generic_mon_call); the close-encounter composite is launched as MON_VERBOSE and stays inline
                bool maybe = false;
                if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) {
#pragma clang fp contract(off)
                    maybe = mp.log_interval != 0. && s.t >= ms.next_log_time;
                } else if (mp.ej_any) {
#pragma clang fp contract(off)
#pragma unroll
                    for (int b = 1; b < N; b++) {
                        maybe = maybe || ej_maybe_body(s.p[b], s.v[b], mp);
                    }
                }
                if (__any(maybe)) {
                    State<N> tmp = s;
                    MonState mtmp = ms;
                    generic_mon_call<N, KIND>(&mtmp, &tmp, &mp, &lg);
                    s = tmp;
                    ms = mtmp;
                }
            } else if (MON) {
                const bool needs_std = mon_pass_one<N, MON == MON_VERBOSE>(ms, s, mp, lg);
                if (is_mvs<KIND> && needs_std) mvs_to_std(s);
                mon_pass_two(ms, s, mp);
                const bool want = mon_need_log(ms, mp);
                if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL || mp.ej_log || mp.ce_log)
                    if (__any(want)) log_system(lg, s, want);
                if (is_mvs<KIND> && needs_std) mvs_to_internal(s);
            }
            if ((!WHOLE || live) && s.state == SWARM_HIP_ACTIVE && s.t >= dest) s.state = SWARM_HIP_INACTIVE;
            live = live && s.state == SWARM_HIP_ACTIVE;
        }
        // prop.shutdown()
        if (is_mvs<KIND>) mvs_to_std(s);
        if (mine && !stored) store_state(e, gid, s);
    }
    if (ghost) return;
    if (cnt.accepted) cnt.accepted[gid] += steps;
    if (cnt.attempted) cnt.attempted[gid] += steps;
}

template <int N, int KIND>
static int launch_n(const LaunchCtx &c) {
#define SWH_LAUNCH_VARIANT(V) \
    hipLaunchKernelGGL((generic_kernel<N, KIND, V>), dim3(c.grid), dim3(c.block), 0, c.stream, c.ens, c.ip, c.mp, c.lg, c.cnt)
    switch (monitor_variant(c.mp)) {
    case MON_OFF: SWH_LAUNCH_VARIANT(MON_OFF); break;
    case MON_ON:
        if (c.mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE && c.mp.ce_any)
            SWH_LAUNCH_VARIANT(MON_VERBOSE); // no cheap early-out: inline monitor (prints only if ce_verbose is set)
        else
            SWH_LAUNCH_VARIANT(MON_ON);
        break;
    default: SWH_LAUNCH_VARIANT(MON_VERBOSE); break;
    }
#undef SWH_LAUNCH_VARIANT
    return check_launch("propagator kernel launch");
}

// One lane per body pays while the ensemble is too small to fill the chip with wavefronts of whole systems
// (kernels_mvs_body.hip).  Crossovers measured with WORKLOAD=c4 tools/lane_sweep.sh (profiles/r2/mvs_lane_sweep.txt):
// nsys ~ 8k for nbod 3, 24-36k for nbod 4..10; at 4096 systems the lane-per-body kernel is 1.5x (nbod 4) to 3.9x
// (nbod 10) faster.  Limits sit at the last grid point where it still wins.
static bool small_mvs_ensemble(const LaunchCtx &c) {
    static const int limit[SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM + 1] = {0, 0, 0, 6144, 20480, 24576, 24576, 24576, 32768, 24576, 24576};
    return within_limit(c, limit);
}

// euler, midpoint, verlet, hermite_prop: one lane per body (kernels_prop_body.hip) for small ensembles.  Measured
// crossovers (profiles/r2/prop_lane_sweep.txt): the lane-per-system kernels take the same time from 2048 to 65536
// systems (one wavefront per SIMD at most), the lane-per-body kernels win below ~16384 systems from nbod 4 on (nbod 10,
// 2048 systems: 4-5x); at nbod 3 only hermite_prop gains anything; verlet, whose lanes also add up the system's
// time-step sum, from nbod 5 on.
static bool prop_prefers_body(const LaunchCtx &c, int integrator) {
    const int nbod = c.ens.nbod, nsys = c.effective_nsys();
    if (integrator == SWARM_HIP_VERLET) return nbod >= 5 && nsys <= (nbod < 8 ? 8192 : 16384);
    if (nbod == 3) return integrator == SWARM_HIP_HERMITE_PROP && nsys <= 8192;
    return nsys <= 16384;
}

int launch_generic(const LaunchCtx &c, int integrator) {
    // small ensembles with nothing to monitor, or a monitor that does not print: one lane per body
    const bool body_ok = body_monitor_of(c.mp, c.ens.nbod) != MONB_NONE;
    int layout = 0;
    switch (integrator) {
    case SWARM_HIP_MVS: layout = choose_layout(c, body_ok, small_mvs_ensemble(c)); break;
    case SWARM_HIP_EULER:
    case SWARM_HIP_VERLET:
    case SWARM_HIP_MIDPOINT:
    case SWARM_HIP_HERMITE_PROP: layout = choose_layout(c, body_ok, prop_prefers_body(c, integrator)); break;
    default: break;
    }
    if (layout < 0) return layout;
    if (layout == 1) return integrator == SWARM_HIP_MVS ? launch_mvs_body(c) : launch_prop_body(c, integrator);
    switch (integrator) {
    case SWARM_HIP_MVS:
        if (c.ip.kepler_compensated) {
            SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, KIND_MVS_COMPENSATED>(c)));
        } else {
            SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, SWARM_HIP_MVS>(c)));
        }
        break;
    case SWARM_HIP_EULER:
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, SWARM_HIP_EULER>(c)));
        break;
    case SWARM_HIP_VERLET:
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, SWARM_HIP_VERLET>(c)));
        break;
    case SWARM_HIP_MIDPOINT:
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, SWARM_HIP_MIDPOINT>(c)));
        break;
    case SWARM_HIP_HERMITE_PROP:
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, SWARM_HIP_HERMITE_PROP>(c)));
        break;
    default:
        set_error("unknown propagator id %d", integrator);
        return SWARM_HIP_EINVAL;
    }
    return SWARM_HIP_OK;
}

} // namespace swh
