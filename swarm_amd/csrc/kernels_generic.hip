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
#include "launch.hpp"

namespace swh {

// sin and cos on [0, pi/4] by the fdlibm kernel polynomials (k_sin.c / k_cos.c, < 1 ulp / ~1 ulp): the
// argument of the Stumpff functions is the eccentric-anomaly increment of one step, so it is almost always
// in that range, and the library sincos() with its full range reduction costs three times as much.
__device__ __forceinline__ void sincos_quarter(const double x, double &sn, double &cs) {
    const double z = x * x;
    const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.58969099521155010221e-10,
                      -2.50507602534068634195e-08), 2.75573137070700676789e-06), -1.98412698298579493134e-04),
                      8.33333333332248946124e-03);
    sn = __builtin_fma(x * z, __builtin_fma(z, rs, -1.66666666666666324348e-01), x);
    const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z,
                      -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                      2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
    cs = 1.0 - __builtin_fma(0.5, z, -(z * rc));
}

// ---- universal-variable Kepler drift: propagators/keplerian.hpp ------------------------------
// SC_prussing, keplerian.hpp:61-87, including `float absy` and the float sqrt it selects.
__device__ __forceinline__ void SC_prussing(double y, double &S, double &C) {
    if (fabs(y) < 1e-4) {
        S = (1.0 / 6.0) * (1.0 - y * (1.0 / 20.0) * (1.0 - y * (1.0 / 42.0) * (1.0 - y * (1.0 / 72.0))));
        C = (1.0 / 2.0) * (1.0 - y * (1.0 / 12.0) * (1.0 - y * (1.0 / 30.0) * (1.0 - y * (1.0 / 56.0))));
    } else {
        float absy = (float)fabs(y);
        double u = (double)sqrtf(absy);
        double u3 = u * u * u;
        if (y > 0.0) {
            double sn, cs;
            if (u <= 0.78539816339744830962)
                sincos_quarter(u, sn, cs);
            else
                sincos(u, &sn, &cs);
            S = fdiv(u - sn, u3);
            C = fdiv(1.0 - cs, y);
        } else {
            S = fdiv(sinh(u) - u, u3);
            C = fdiv(cosh(u) - 1.0, -y);
        }
    }
}

// Kepler drift of G planets in lockstep: per planet the iteration of solvex and the f, g update of drift_kepler
// (keplerian.hpp:120-159, 175-236), with the Laguerre iterations of the G planets advancing together so that their G
// independent dependency chains interleave (one lane owns the whole system, so without this every iteration is one
// long serial chain of FP64 latencies).  A planet whose own loop condition `(i < 7) && !((i > 2) && (x + u == x))` has
// failed keeps its x (select, no branch).  The Stumpff functions take the series branch of SC_prussing for all planets
// with one wave-uniform test when every lane is in that regime (|alpha x^2| < 1e-4, the usual case for dt << orbital
// period), and the general per-planet routine otherwise.
//
// The expressions are the reference's, regrouped so that one pass is ~45 issue slots instead of ~70 (the solver is
// three quarters of an MVS step):
//   S, C series       Horner in y with the expanded coefficients 1/3!, -1/5!, 1/7!, -1/9! and 1/2!, -1/4!, 1/6!, -1/8!
//                     (keplerian.hpp:65-66 nests the same polynomial as products of (1 - y/k (...)))
//   F                 x (sig0 xC + foo x^2 S + r0) - smu dt            (L146: sig0 x^2 C + foo x^3 S + r0 x - smu dt)
//   dF                (sig0 x)(1 - yS) + foo x (xC) + r0               (L147)
//   ddF               sig0 (1 - yC) + foo (x (1 - yS))                 (L148)
//   z                 |16 dF^2 - 20 F ddF|                             (L149 with n = 5 written out)
//   r0, 1/r0          one rsqrt of |p|^2 gives both
// each value within an ulp or two of the reference's association; what the tests hold is 1e-10 against the CPU
// restatement (and 4e-12 at MVS.cfg's step, tests/test_gpu_integrators.py).
__device__ __forceinline__ void SC_series(const double y, double &S, double &C) {
    S = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, -1.0 / 362880.0, 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
    C = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, -1.0 / 40320.0, 1.0 / 720.0), -1.0 / 24.0), 0.5);
}
// The general routine, out of line and marked cold.  Inlined (14 copies of sincos/sinh/cosh at nbod 8) its register
// needs put ~55 spill stores and reloads per step on the hot path of a kernel that never takes it -- 7.2 GB of scratch
// traffic per C4-shard launch; measured on that shard (series regime): inline 14.3 ms, inline behind a branch-weight
// hint 12.2 ms, this 10.8 ms, no general path at all 10.3 ms; in the trigonometric regime (systems shrunk 30x) 37.7 ms
// inline against 19.1 ms.
static __device__ __attribute__((noinline, cold)) void SC_prussing_outlined(double y, double *sc) {
    double S, C;
    SC_prussing(y, S, C);
    sc[0] = S, sc[1] = C;
}
template <int G>
__device__ __forceinline__ void SC_group(const double (&y)[G], double (&S)[G], double (&C)[G]) {
    bool small = true;
#pragma unroll
    for (int g = 0; g < G; g++) small = small && (fabs(y[g]) < 1e-4);
    if (__builtin_expect(__all(small), 1)) {
#pragma unroll
        for (int g = 0; g < G; g++) SC_series(y[g], S[g], C[g]);
    } else {
#ifndef SWH_EXPERIMENT_SERIES_ONLY /* ceiling experiment: no general Stumpff path at all; never in the shipped build */
#pragma unroll
        for (int g = 0; g < G; g++) {
            double sc[2];
            SC_prussing_outlined(y[g], sc);
            S[g] = sc[0], C[g] = sc[1];
        }
#endif
    }
}

// what the solver needs of the star and of the step; formed when the step size changes (once per launch, and again
// for a clipped last step)
struct KeplerConsts {
    double smu, inv_smu, inv_GM; // sqrt(GM) of the star, its reciprocal, 1/GM
    double dt, smudt, gmdt2;     // the drift's time span, smu dt, GM dt^2 (the numerator of solvex's initial guess)
    __device__ __forceinline__ void set_star(const double sqrtGM) {
        smu = sqrtGM;
        inv_smu = rcp_full(sqrtGM);
        inv_GM = rcp_full(sqrtGM * sqrtGM);
        dt = -1.0;
    }
    __device__ __forceinline__ void set_dt(const double deltaTime) {
        dt = deltaTime;
        smudt = smu * deltaTime;
        gmdt2 = smu * smu * deltaTime * deltaTime;
    }
};

template <int N, int B0, int G>
__device__ __forceinline__ void drift_kepler_group(State<N> &s, const KeplerConsts &k, const bool compensated) {
    double r0[G], inv_r0[G], alpha[G], sig0[G], foo[G], x[G], u[G];
    bool live[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const double(&p)[3] = s.p[B0 + g];
        const double(&v)[3] = s.v[B0 + g];
        const double rr = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        inv_r0[g] = rsqrt_full(rr);
        r0[g] = rr * inv_r0[g];
        const double v2 = (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double r0dotv0 = (p[0] * v[0] + p[1] * v[1] + p[2] * v[2]);
        alpha[g] = (2.0 * inv_r0[g] - v2 * k.inv_GM);
        foo[g] = 1.0 - r0[g] * alpha[g];
        sig0[g] = r0dotv0 * k.inv_smu;
        x[g] = k.gmdt2 * inv_r0[g];
        u[g] = 1.0;
        live[g] = true;
    }
    for (int i = 0; i < 7; i++) {
        bool any = false;
#pragma unroll
        for (int g = 0; g < G; g++) {
            live[g] = live[g] && !((i > 2) && (x[g] + u[g] == x[g]));
            any = any || live[g];
        }
        if (!__any(any)) break;
        double y[G], Sp[G], Cp[G];
#pragma unroll
        for (int g = 0; g < G; g++) y[g] = alpha[g] * (x[g] * x[g]);
        SC_group<G>(y, Sp, Cp);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const double xx = x[g], x2 = xx * xx;
            const double a = __builtin_fma(-y[g], Sp[g], 1.0), b = __builtin_fma(-y[g], Cp[g], 1.0);
            const double xC = xx * Cp[g], x2S = x2 * Sp[g];
            const double F = __builtin_fma(xx, __builtin_fma(sig0[g], xC, __builtin_fma(foo[g], x2S, r0[g])), -k.smudt);
            const double dF = __builtin_fma(sig0[g] * xx, a, __builtin_fma(foo[g], xx * xC, r0[g]));
            const double ddF = __builtin_fma(sig0[g], b, foo[g] * (xx * a));
            const double az = fabs(__builtin_fma(16.0 * dF, dF, -20.0 * (F * ddF)));
            const double z = az * rsqrt_full(fmax(az, 1e-300)); // sqrt; 0 for az = 0
            double denom = dF < 0 ? dF - z : dF + z;
            if (denom == 0.0) denom = 1e-8;
            const double un = (5.0 * F) * rcp_full(denom);
            u[g] = live[g] ? un : u[g];
            x[g] = live[g] ? xx - un : xx;
        }
    }
    double y[G], Sp[G], Cp[G];
#pragma unroll
    for (int g = 0; g < G; g++) y[g] = alpha[g] * (x[g] * x[g]);
    SC_group<G>(y, Sp, Cp);
#pragma unroll
    for (int g = 0; g < G; g++) {
        double(&p)[3] = s.p[B0 + g];
        double(&v)[3] = s.v[B0 + g];
        const double x_old = p[0], y_old = p[1], z_old = p[2], vx_old = v[0], vy_old = v[1], vz_old = v[2];
        const double x_p = x[g], x2 = x_p * x_p;
        const double a = __builtin_fma(-y[g], Sp[g], 1.0);
        const double x2C = x2 * Cp[g];
        double r = __builtin_fma(sig0[g] * x_p, a, __builtin_fma(foo[g], x2C, r0[g]));
        if (r < 0.) r = 0.;
        const double g_p = __builtin_fma(-(x2 * x_p) * k.inv_smu, Sp[g], k.dt);
        // fh = f - 1 and gh = dg/dt - 1 (keplerian.hpp:214, 218)
        const double fh = -(x2C * inv_r0[g]);
        const double gh = -(x2C * rcp_full(r));
        double f_p, dgdt, num;
        if (!compensated) { // keplerian.hpp:214-231 as written: f and dg/dt rounded to doubles next to 1
            f_p = 1.0 + fh;
            dgdt = 1.0 + gh;
            num = __builtin_fma(f_p, dgdt, -1.0);
        } else {
            // The same expressions in terms of fh and gh: f*dgdt - 1 = fh + gh + fh*gh.  Written with f and dgdt, both
            // within ~GM dt^2/r^3 of 1, the difference f*dgdt - 1 keeps only the digits of fh and gh above the spacing
            // of the doubles at 1 (1.1e-16), an error of ~1e-16/dt in dfdt that repeats with the same sign step after
            // step (MVS.cfg, dt = 3e-4: 6e-9 ... 9e-8 in velocity at T = 1 for nbod 3..8, tools/mvs_parity.py).
            // swarm_hip_launch_t::kepler_compensated, not a reference key.
            num = fh + gh + fh * gh;
        }
        double dfdt = num * rcp_full(g_p);
        if (__builtin_expect(__any(!(fabs(g_p) > 1e-8)), 0)) { // L222-224, a step too short for the quotient
            const double alt = fdiv(x_p * k.smu, r * r0[g]) * (y[g] * Sp[g] - 1.0);
            dfdt = fabs(g_p) > 1e-8 ? dfdt : alt;
        }
        if (!compensated) {
            p[0] = f_p * x_old + g_p * vx_old;
            p[1] = f_p * y_old + g_p * vy_old;
            p[2] = f_p * z_old + g_p * vz_old;
            v[0] = dfdt * x_old + dgdt * vx_old;
            v[1] = dfdt * y_old + dgdt * vy_old;
            v[2] = dfdt * z_old + dgdt * vz_old;
        } else {
            p[0] = x_old + (fh * x_old + g_p * vx_old);
            p[1] = y_old + (fh * y_old + g_p * vy_old);
            p[2] = z_old + (fh * z_old + g_p * vz_old);
            v[0] = vx_old + (dfdt * x_old + gh * vx_old);
            v[1] = vy_old + (dfdt * y_old + gh * vy_old);
            v[2] = vz_old + (dfdt * z_old + gh * vz_old);
        }
    }
}

// all planets 1..N-1, in groups of at most KEPLER_GROUP (register budget: 6 doubles per planet)
#ifndef SWH_KEPLER_GROUP
#define SWH_KEPLER_GROUP 2 // measured on MI355X, C4 shard (ms per 1001 steps): G=1 20.7, 2 18.5, 4 19.7, 7 20.9
#endif
constexpr int KEPLER_GROUP = SWH_KEPLER_GROUP;
template <int N, int B0 = 1>
__device__ __forceinline__ void drift_kepler_all(State<N> &s, const KeplerConsts &k, const bool compensated) {
    if constexpr (B0 < N) {
        constexpr int G = (N - B0 < KEPLER_GROUP) ? (N - B0) : KEPLER_GROUP;
        drift_kepler_group<N, B0, G>(s, k, compensated);
        drift_kepler_all<N, B0 + G>(s, k, compensated);
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
            sump += mj * s.p[j][c];
            sumv += mj * s.v[j][c];
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
            sump += mj * s.p[j][c];
            sumv += mj * s.v[j][c];
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
template <int N>
__device__ __forceinline__ void mvs_drift(State<N> &s, double hby2, double inv_m0) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double mv = 0;
#pragma unroll
        for (int j = 1; j < N; ++j) mv += s.m[j] * s.v[j][c];
        s.p[0][c] += hby2 * s.v[0][c];
#pragma unroll
        for (int b = 1; b < N; b++) s.p[b][c] += mv * hby2 * inv_m0;
    }
}

// Verlet time step: VerletPropagator::calc_timestep (verlet.hpp:112-137) over
// GravitationAcc::one_over_r (gravitation_acc.hpp:78-93), which returns |dx r^-3|^2 = r^-4.
template <int N>
__device__ __forceinline__ double verlet_timestep(const State<N> &s, const IntParams &ip) {
    double factor = 0.;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 1; j < N; ++j) {
            if (i == j) continue;
            const double dx0 = s.p[j][0] - s.p[i][0], dx1 = s.p[j][1] - s.p[i][1], dx2 = s.p[j][2] - s.p[i][2];
            const double r2 = dx0 * dx0 + dx1 * dx1 + dx2 * dx2;
            const double y = rsqrt_full(r2);
            const double am = y * (y * y);
            double oor = 0.;
            oor += (dx0 * am) * (dx0 * am);
            oor += (dx1 * am) * (dx1 * am);
            oor += (dx2 * am) * (dx2 * am);
            factor += (s.m[i] + s.m[j]) * oor * oor * oor;
        }
    }
    factor *= ip.timestep_scale * ip.timestep_scale;
    factor += 1.;
    factor = rsqrt_full(factor);
    factor *= ip.max_timestep;
    return factor;
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
    if (KIND == SWARM_HIP_MVS && needs_std) mvs_to_std(s);
    mon_pass_two(*ms, s, mp);
    const bool want = mon_need_log(*ms, mp);
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL || mp.ej_log || mp.ce_log)
        if (__any(want)) log_system(*lgp, s, want);
    if (KIND == SWARM_HIP_MVS && needs_std) mvs_to_internal(s);
}

// MVS keeps two waves per SIMD (<= 256 registers, a few spilled doubles from nbod 5 on): its Kepler solver is a
// chain of dependent FP64 operations with quarter-rate reciprocals, and a second resident wave hides them
// (131072 x 5..8: -15..20 %; at one wave of work per SIMD, 65536 systems, +2..3 %).
template <int N, int KIND, int MON>
__global__ __launch_bounds__(256, KIND == SWARM_HIP_MVS ? 2 : 1) void generic_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
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
    if (WHOLE ? __any(mine) : mine) {
        bool live = mine;
        MonState ms;
        mon_init(ms, s);
        double acc[N][3];
        double timestep = 0.;
        const double inv_m0 = 1.0 / s.m[0];
        KeplerConsts kc;
        // prop.init()
        if (KIND == SWARM_HIP_MVS) {
            kc.set_star(sqrt(s.m[0]));
            mvs_to_internal(s);
            acc_only<N, true>(s.p, s.m, acc);
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

        for (int iter = 0; iter < ip.max_iterations && (WHOLE ? __any(live) : live); iter++) {
            double max_timestep;
            {
#pragma clang fp contract(off)
                max_timestep = dest - s.t;
            }
            if (KIND == SWARM_HIP_MVS) { // MVSPropagator::advance, mvs.hpp:260-293
                const double hby2 = (!WHOLE || live) ? 0.5 * fmin(max_timestep, ip.time_step) : 0.0;
                mvs_drift(s, hby2, inv_m0);
#pragma unroll
                for (int b = 1; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) s.v[b][c] += hby2 * acc[b][c];
                if (2.0 * hby2 != kc.dt) kc.set_dt(2.0 * hby2);
                drift_kepler_all<N>(s, kc, ip.kepler_compensated != 0);
                acc_only<N, true>(s.p, s.m, acc);
#pragma unroll
                for (int b = 1; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) s.v[b][c] += hby2 * acc[b][c];
                mvs_drift(s, hby2, inv_m0);
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
                const double third = 1.0 / 3.0;
                double jerk[N][3];
                accjerk<N>(s.p, s.v, s.m, acc, jerk);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const double pos = s.p[b][c], vel = s.v[b][c];
                        s.p[b][c] = pos + h * (vel + (h * 0.5) * (acc[b][c] + (h * third) * jerk[b][c]));
                        s.v[b][c] = vel + h * (acc[b][c] + (h * 0.5) * jerk[b][c]);
                    }
                s.t += h;
            } else if (KIND == SWARM_HIP_VERLET) { // VerletPropagator::advance, verlet.hpp:139-175
                const double h = (!WHOLE || live) ? timestep : 0.0;
                const double h_first_half = 0.5 * h;
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) s.p[b][c] = s.p[b][c] + h_first_half * s.v[b][c];
                acc_only<N, false>(s.p, s.m, acc);
                timestep = verlet_timestep(s, ip);
                const double h_second_half = (!WHOLE || live) ? 0.5 * timestep : 0.0;
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        double vel = s.v[b][c] + h_first_half * acc[b][c];
                        vel = vel + h_second_half * acc[b][c];
                        s.v[b][c] = vel;
                        s.p[b][c] = s.p[b][c] + h_second_half * vel;
                    }
                s.t += h_first_half + h_second_half;
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
                        p_i[b][c] = p_im1[b][c] + h * v_im1[b][c];
                        v_i[b][c] = v_im1[b][c] + h * acc[b][c];
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
                            p_i[b][c] = p_im2[b][c] + 2.0 * h * v_im1[b][c];
                            v_i[b][c] = v_im2[b][c] + 2.0 * h * acc[b][c];
                        }
                }
                acc_only<N, false>(p_i, s.m, acc);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        s.p[b][c] = 0.5 * (p_i[b][c] + p_im1[b][c] + h * v_i[b][c]);
                        s.v[b][c] = 0.5 * (v_i[b][c] + v_im1[b][c] + h * acc[b][c]);
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
                        const double r2 = s.p[b][0] * s.p[b][0] + s.p[b][1] * s.p[b][1] + s.p[b][2] * s.p[b][2];
                        maybe = maybe || !(r2 < mp.r2_below);
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
                if (KIND == SWARM_HIP_MVS && needs_std) mvs_to_std(s);
                mon_pass_two(ms, s, mp);
                const bool want = mon_need_log(ms, mp);
                if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL || mp.ej_log || mp.ce_log)
                    if (__any(want)) log_system(lg, s, want);
                if (KIND == SWARM_HIP_MVS && needs_std) mvs_to_internal(s);
            }
            if ((!WHOLE || live) && s.state == SWARM_HIP_ACTIVE && s.t >= dest) s.state = SWARM_HIP_INACTIVE;
            live = live && s.state == SWARM_HIP_ACTIVE;
        }
        // prop.shutdown()
        if (KIND == SWARM_HIP_MVS) mvs_to_std(s);
        if (mine) store_state(e, gid, s);
    }
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

int launch_generic(const LaunchCtx &c, int integrator) {
    switch (integrator) {
    case SWARM_HIP_MVS:
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, SWARM_HIP_MVS>(c)));
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
