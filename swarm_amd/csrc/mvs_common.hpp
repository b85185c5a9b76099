// mvs_common.hpp -- the universal-variable Kepler drift of one planet (propagators/keplerian.hpp), shared by the two
// MVS kernels: one lane per system (kernels_generic.hip, planets in groups) and one lane per body
// (kernels_mvs_body.hip).  Every floating-point operation is written out (explicit fma, contraction off), so both
// kernels perform the same operations on the same operands and give bit-identical ensembles
// (tests/test_gpu_mvs_body.py).
#pragma once
#include "device_common.hpp"

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

// ---- Stumpff functions: SC_prussing, keplerian.hpp:61-87 -------------------------------------
// Series branch (|y| < 1e-4): Horner in y with the expanded coefficients 1/3!, -1/5!, 1/7!, -1/9! and 1/2!, -1/4!,
// 1/6!, -1/8! (keplerian.hpp:65-66 nests the same polynomial as products of (1 - y/k (...))).
__device__ __forceinline__ void SC_series(const double y, double &S, double &C) {
    S = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, -1.0 / 362880.0, 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
    C = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, -1.0 / 40320.0, 1.0 / 720.0), -1.0 / 24.0), 0.5);
}
// The whole routine, including `float absy` and the float sqrt it selects.  Its series branch is SC_series, so what a
// planet gets does not depend on which of the two routes (wave-uniform series, or this) its wavefront took.
__device__ __forceinline__ void SC_prussing(double y, double &S, double &C) {
    if (fabs(y) < 1e-4) {
        SC_series(y, S, C);
    } else {
#pragma clang fp contract(off)
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

// ---- one planet's drift along its Kepler orbit: solvex and drift_kepler, keplerian.hpp:120-159, 175-236 ----------
// The caller owns the iteration (so that the chains of several planets, or of 64 lanes, advance together):
//     kepler_begin;  for (i = 0; i < 7; i++) { if (no planet passes kepler_more(i)) break;
//                                              y = kepler_y; S, C = Stumpff(y); kepler_pass; }
//     y = kepler_y; S, C = Stumpff(y); kepler_end
// A planet whose own loop condition `(i < 7) && !((i > 2) && (x + u == x))` has failed keeps its x (select, no branch).
//
// The expressions are the reference's, regrouped so that one pass is ~45 issue slots instead of ~70 (the solver is
// three quarters of an MVS step):
//   F                 x (sig0 xC + foo x^2 S + r0) - smu dt            (L146: sig0 x^2 C + foo x^3 S + r0 x - smu dt)
//   dF                (sig0 x)(1 - yS) + foo x (xC) + r0               (L147)
//   ddF               sig0 (1 - yC) + foo (x (1 - yS))                 (L148)
//   z                 |16 dF^2 - 20 F ddF|                             (L149 with n = 5 written out)
//   r0, 1/r0          one rsqrt of |p|^2 gives both
// each value within an ulp or two of the reference's association; what the tests hold is 1e-10 against the CPU
// restatement (and 4e-12 at MVS.cfg's step, tests/test_gpu_integrators.py).
struct KeplerOrbit {
    double r0, inv_r0, alpha, sig0, foo; // |p|, 1/|p|, 2/r0 - v^2/GM, p.v/sqrt(GM), 1 - r0 alpha
    double x, u;                         // the universal variable and its last correction
    bool live;                           // still iterating
};

__device__ __forceinline__ void kepler_begin(KeplerOrbit &o, const double (&p)[3], const double (&v)[3],
                                             const KeplerConsts &k) {
#pragma clang fp contract(off)
    const double rr = __builtin_fma(p[2], p[2], __builtin_fma(p[1], p[1], p[0] * p[0]));
    o.inv_r0 = rsqrt_full(rr);
    o.r0 = rr * o.inv_r0;
    const double v2 = __builtin_fma(v[2], v[2], __builtin_fma(v[1], v[1], v[0] * v[0]));
    const double r0dotv0 = __builtin_fma(p[2], v[2], __builtin_fma(p[1], v[1], p[0] * v[0]));
    o.alpha = __builtin_fma(-v2, k.inv_GM, 2.0 * o.inv_r0);
    o.foo = __builtin_fma(-o.r0, o.alpha, 1.0);
    o.sig0 = r0dotv0 * k.inv_smu;
    o.x = k.gmdt2 * o.inv_r0;
    o.u = 1.0;
    o.live = true;
}
__device__ __forceinline__ bool kepler_more(KeplerOrbit &o, const int i) {
#pragma clang fp contract(off)
    o.live = o.live && !((i > 2) && (o.x + o.u == o.x));
    return o.live;
}
__device__ __forceinline__ double kepler_y(const KeplerOrbit &o) {
#pragma clang fp contract(off)
    return o.alpha * (o.x * o.x);
}
__device__ __forceinline__ void kepler_pass(KeplerOrbit &o, const KeplerConsts &k, const double y, const double S,
                                            const double C) {
#pragma clang fp contract(off)
    const double xx = o.x, x2 = xx * xx;
    const double a = __builtin_fma(-y, S, 1.0), b = __builtin_fma(-y, C, 1.0);
    const double xC = xx * C, x2S = x2 * S;
    const double F = __builtin_fma(xx, __builtin_fma(o.sig0, xC, __builtin_fma(o.foo, x2S, o.r0)), -k.smudt);
    const double dF = __builtin_fma(o.sig0 * xx, a, __builtin_fma(o.foo, xx * xC, o.r0));
    const double ddF = __builtin_fma(o.sig0, b, o.foo * (xx * a));
    const double az = fabs(__builtin_fma(16.0 * dF, dF, -20.0 * (F * ddF)));
    const double z = az * rsqrt_full(fmax(az, 1e-300)); // sqrt; 0 for az = 0
    // dF + SIGN(dF) z, keplerian.hpp:150-151, as one sign transfer and one addition: `dF < 0 ? dF - z : dF + z` differs
    // from dF + copysign(z, dF) only for dF = -0.0, which the sum above cannot produce (a cancellation to zero gives
    // +0.0 in round-to-nearest, and its addend r0 is positive), and both give NaN for a NaN
    double denom = dF + __builtin_copysign(z, dF);
    if (denom == 0.0) denom = 1e-8;
    const double f5 = 5.0 * F, rd = rcp_full(denom);
    const double un = f5 * rd; // the correction, kept for the convergence test; x - un in one rounding, off un's chain
    o.u = o.live ? un : o.u;
    o.x = o.live ? __builtin_fma(-f5, rd, xx) : xx;
}
// `counts`: whether this lane's planet is a real one (the lane-per-body kernel carries the star's and idle lanes
// through the same instructions); only real planets can send the wavefront through the short-step branch.
__device__ __forceinline__ void kepler_end(const KeplerOrbit &o, const KeplerConsts &k, const double y, const double S,
                                           const double C, const bool compensated, const bool counts, double (&p)[3],
                                           double (&v)[3]) {
#pragma clang fp contract(off)
    const double x_p = o.x, x2 = x_p * x_p;
    const double a = __builtin_fma(-y, S, 1.0);
    const double x2C = x2 * C;
    // `if (r < 0) r = 0`, keplerian.hpp:210-211, as one maximum (a NaN, which the comparison keeps and the maximum
    // turns into 0, ends as NaN in the reciprocal below either way; -0.0 cannot come out of a sum with r0 > 0)
    const double r = fmax(__builtin_fma(o.sig0 * x_p, a, __builtin_fma(o.foo, x2C, o.r0)), 0.);
    const double g_p = __builtin_fma(-(x2 * x_p) * k.inv_smu, S, k.dt);
    // fh = f - 1 and gh = dg/dt - 1 (keplerian.hpp:214, 218)
    const double fh = -(x2C * o.inv_r0);
    const double gh = -(x2C * rcp_full(r));
    double f_p = 0., dgdt = 0., num;
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
        num = __builtin_fma(fh, gh, fh + gh);
    }
    double dfdt = num * rcp_full(g_p);
    if (__builtin_expect(__any(counts && !(fabs(g_p) > 1e-8)), 0)) { // L222-224, a step too short for the quotient
        const double alt = fdiv(x_p * k.smu, r * o.r0) * __builtin_fma(y, S, -1.0);
        dfdt = fabs(g_p) > 1e-8 ? dfdt : alt;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double q = p[c], w = v[c];
        if (!compensated) {
            p[c] = __builtin_fma(f_p, q, g_p * w);
            v[c] = __builtin_fma(dfdt, q, dgdt * w);
        } else {
            p[c] = q + __builtin_fma(fh, q, g_p * w);
            v[c] = w + __builtin_fma(dfdt, q, gh * w);
        }
    }
}

} // namespace swh
