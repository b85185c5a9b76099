// rkck_common.hpp -- the per-coordinate, per-body and per-system arithmetic of the Cash-Karp integrator
// (integrators/rkck_integrator.hpp:90-294), shared by the two kernels: one lane per system (kernels_rkck.hip) and one
// lane per body (kernels_rkck_body.hip).  Everything of the method except the forces acts on one coordinate of one
// body across the stages (stage positions, solution, error estimate), on one body (its normalised error) or on one
// system (step control); written once here with every floating-point operation explicit, so that both kernels
// produce the same bits.
#pragma once
#include "device_common.hpp"

namespace swh {

// Cash-Karp tableau, rkck_integrator.hpp:96-108 (row k = coefficients of stage k+1)
struct CK {
    static constexpr double b[6][6] = {
        {0, 0, 0, 0, 0, 0},
        {1.0 / 5.0, 0, 0, 0, 0, 0},
        {3.0 / 40.0, 9.0 / 40.0, 0, 0, 0, 0},
        {0.3, -0.9, 1.2, 0, 0, 0},
        {-11.0 / 54.0, 2.5, -70.0 / 27.0, 35.0 / 27.0, 0, 0},
        {1631.0 / 55296.0, 175.0 / 512.0, 575.0 / 13824.0, 44275.0 / 110592.0, 253.0 / 4096.0, 0}};
    static constexpr double b6[6] = {37.0 / 378.0, 0, 250.0 / 621.0, 125.0 / 594.0, 0, 512.0 / 1771.0};
    static constexpr double ecc[6] = {37.0 / 378.0 - 2825.0 / 27648.0, 0.0, 250.0 / 621.0 - 18575.0 / 48384.0,
                                      125.0 / 594.0 - 13525.0 / 55296.0, -277.00 / 14336.0, 512.0 / 1771.0 - 0.25};
    // weights w (length 6) over stage velocities k_vel[i] = vel + h * sum_j b[i][j] k_acc[j]:
    //   sum_i w[i] k_vel[i] = rowsum(w) * vel + h * sum_j second(w, j) * k_acc[j]
    static constexpr double rowsum(const double (&w)[6], int n) {
        double s = 0;
        for (int i = 0; i < n; i++) s += w[i];
        return s;
    }
    static constexpr double second(const double (&w)[6], int n, int j) {
        double s = 0;
        for (int i = 0; i < n; i++) s += w[i] * b[i][j];
        return s;
    }
};

// The reference keeps k_vel[1..6] and k_acc[1..6] per (body,component) in registers of 3*nbod
// threads.  With one lane per system that is 12 * 3 * nbod doubles per lane (nbod=6: 432 VGPRs)
// before any temporaries.  The stage velocities are linear in vel and k_acc, so only k_acc is
// kept and every stage position is formed as
//     p_k = pos + h * (c_k * vel + h * sum_j B2[k][j] * k_acc[j]),  B2 = b * b, c_k = rowsum(b[k])
// which is the same polynomial as the reference's `pos + h*(b[k][0]*k1_vel + ...)` (L158-194)
// with the products of tableau constants folded at compile time; evaluated as
//     p_k = fma(h^2, sum_j B2[k][j] k_acc[j], fma(c_k h, vel, pos))
// (one operation per component and stage fewer than h (c_k vel + h sum): C3 8.33 -> 8.04 ms).

// position of one coordinate at stage K+1 (K = 0..4) from the stage accelerations ka[0..K-1]
template <int K>
__device__ __forceinline__ double rkck_stage_position(const double h, const double p, const double v,
                                                      const double (&ka)[6]) {
#pragma clang fp contract(off)
    constexpr double ck = CK::rowsum(CK::b[K + 1], 6);
    double acc2 = 0.0;
#pragma unroll
    for (int j = 0; j < K; j++) { // second(b[K+1], j) is zero for j >= K
        const double w = CK::second(CK::b[K + 1], 6, j);
        acc2 = (j == 0) ? w * ka[j] : __builtin_fma(w, ka[j], acc2);
    }
    // pos + (c_k h) vel + h^2 sum: one operation per component fewer than h (c_k vel + h sum)
    const double lin = __builtin_fma(ck * h, v, p);
    return (K == 0) ? lin : __builtin_fma(h * h, acc2, lin);
}

// 5th-order solution of one coordinate and, ADAPTIVE, its embedded error estimates (L193-199)
template <bool ADAPTIVE>
__device__ __forceinline__ void rkck_solution(const double h, const double p, const double v, const double (&ka)[6],
                                              double &p6, double &v6, double &pe, double &ve) {
#pragma clang fp contract(off)
    double sa = 0, sa2 = 0, ea = 0, ea2 = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const double kj = ka[j];
        sa = (j == 0) ? CK::b6[0] * kj : __builtin_fma(CK::b6[j], kj, sa);
        sa2 = (j == 0) ? CK::second(CK::b6, 6, 0) * kj : __builtin_fma(CK::second(CK::b6, 6, j), kj, sa2);
        if (ADAPTIVE) {
            ea = (j == 0) ? CK::ecc[0] * kj : __builtin_fma(CK::ecc[j], kj, ea);
            ea2 = (j == 0) ? CK::second(CK::ecc, 6, 0) * kj : __builtin_fma(CK::second(CK::ecc, 6, j), kj, ea2);
        }
    }
    v6 = __builtin_fma(h, sa, v);
    p6 = __builtin_fma(h * h, sa2, __builtin_fma(CK::rowsum(CK::b6, 6) * h, v, p));
    if (ADAPTIVE) {
        ve = h * ea;
        pe = __builtin_fma(h * h, ea2, (CK::rowsum(CK::ecc, 6) * h) * v);
    }
}

// normalised error of one body, L232-238: |error|^2 / |solution|^2 for position and velocity, the larger of the two;
// quotients through the 1-ulp reciprocal path: the estimate itself is a difference of nearly equal sums and is good
// to a few digits only
__device__ __forceinline__ double rkck_body_error(const double (&pe)[3], const double (&ve)[3], const double (&p6)[3],
                                                  const double (&v6)[3]) {
#pragma clang fp contract(off)
    const double pem = __builtin_fma(pe[2], pe[2], __builtin_fma(pe[1], pe[1], pe[0] * pe[0]));
    const double vem = __builtin_fma(ve[2], ve[2], __builtin_fma(ve[1], ve[1], ve[0] * ve[0]));
    const double pm = __builtin_fma(p6[2], p6[2], __builtin_fma(p6[1], p6[1], p6[0] * p6[0]));
    const double vm = __builtin_fma(v6[2], v6[2], __builtin_fma(v6[1], v6[1], v6[0] * v6[0]));
    return fmax(fdiv(pem, pm), fdiv(vem, vm));
}

// step control of one system, L204-269: float constants, pow() on the normalised error; updates time_step and
// returns whether the attempt is accepted
__device__ __forceinline__ bool rkck_step_control(const double max_error, double &time_step, const IntParams &ip) {
#pragma clang fp contract(off)
    const float step_grow_power = -1. / (5 + 1.);
    const float step_shrink_power = -1. / 5;
    const float step_guess_safety_factor = 0.9;
    const float step_grow_max_factor = 5.0;
    const float step_shrink_min_factor = 0.2;
    const double normalized_error = max_error / ip.error_tolerance;
    // Cruising at the cap: for normalized_error < 0.25 the grow factor 0.9f * pow(e, -0.0833..) lies in
    // [1.0102, inf] (e = 0 gives inf, clamped to 5), so time_step * factor >= time_step * 1.01 and, once
    // that reaches max_time_step, the fmin below returns max_time_step whatever pow() says: the same value
    // without the 226 instructions of pow().  Waves whose lanes all cruise skip them (C3: every step).
    const bool capped = (normalized_error < 0.25) && (time_step * 1.01 >= ip.max_time_step);
    double new_time_step = ip.max_time_step;
    if (!capped) {
        const double step_guess_power = (normalized_error < 1.) ? step_grow_power : step_shrink_power;
        const double step_change_factor = ((normalized_error < 0.5) || (normalized_error > 1.0))
                                              ? step_guess_safety_factor * pow(normalized_error, 0.5 * step_guess_power)
                                              : 1.0;
        new_time_step = (normalized_error > 1.)
                            ? fmax(time_step * fmax(step_change_factor, (double)step_shrink_min_factor), ip.min_time_step)
                            : fmin(time_step * fmax(fmin(step_change_factor, (double)step_grow_max_factor), 1.0),
                                   ip.max_time_step);
    }
    const bool accept_step = (normalized_error < 1.0) || (fabs(time_step - new_time_step) < 1e-10);
    time_step = new_time_step;
    return accept_step;
}

} // namespace swh
