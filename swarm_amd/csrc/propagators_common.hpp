// propagators_common.hpp -- the per-coordinate arithmetic of the tutorial propagators (propagators/euler.hpp:101-122,
// midpoint.hpp:106-160, verlet.hpp:112-175), shared by the lane-per-system kernel (kernels_generic.hip) and the lane-per-body kernel
// (kernels_prop_body.hip): every floating-point operation explicit, so both produce the same bits.
#pragma once
#include "device_common.hpp"

namespace swh {

// EulerPropagator::advance: third-order Taylor step of one coordinate from acceleration and jerk
//   pos + h (vel + h/2 (acc + h/3 jerk)),  vel + h (acc + h/2 jerk)
__device__ __forceinline__ void euler_advance(const double h, const double acc, const double jerk, double &p, double &v) {
#pragma clang fp contract(off)
    const double third = 1.0 / 3.0;
    p = __builtin_fma(h, __builtin_fma(h * 0.5, __builtin_fma(h * third, jerk, acc), v), p);
    v = __builtin_fma(h, __builtin_fma(h * 0.5, jerk, acc), v);
}

// MidpointPropagator::advance, modified midpoint with n = 4 substeps of h = H / 4, one coordinate:
// the first substep from (p0, v0), ...
__device__ __forceinline__ void midpoint_first(const double h, const double p0, const double v0, const double acc,
                                               double &p_i, double &v_i) {
#pragma clang fp contract(off)
    p_i = __builtin_fma(h, v0, p0);
    v_i = __builtin_fma(h, acc, v0);
}
// ... the leapfrog substeps z_i = z_{i-2} + 2 h f(z_{i-1}), ...
__device__ __forceinline__ void midpoint_next(const double h, const double p_im2, const double v_im2, const double v_im1,
                                              const double acc, double &p_i, double &v_i) {
#pragma clang fp contract(off)
    p_i = __builtin_fma(2.0 * h, v_im1, p_im2);
    v_i = __builtin_fma(2.0 * h, acc, v_im2);
}
// ... and the closing average 1/2 (z_n + z_{n-1} + h f(z_n))
__device__ __forceinline__ void midpoint_last(const double h, const double p_i, const double p_im1, const double v_i,
                                              const double v_im1, const double acc, double &p, double &v) {
#pragma clang fp contract(off)
    p = 0.5 * __builtin_fma(h, v_i, p_i + p_im1);
    v = 0.5 * __builtin_fma(h, acc, v_i + v_im1);
}

// VerletPropagator::calc_timestep (verlet.hpp:112-137): the term of one ordered pair (i, j), (m_i + m_j) u^3 with
// u = |d r^-3|^2 as GravitationAcc::one_over_r forms it (gravitation_acc.hpp:78-93); even in the sign of d
__device__ __forceinline__ double verlet_pair_term(const double d0, const double d1, const double d2, const double msum) {
#pragma clang fp contract(off)
    const double r2 = __builtin_fma(d2, d2, __builtin_fma(d1, d1, d0 * d0));
    const double y = rsqrt_full(r2);
    const double am = y * (y * y);
    const double a0 = d0 * am, a1 = d1 * am, a2 = d2 * am;
    const double u = __builtin_fma(a2, a2, __builtin_fma(a1, a1, a0 * a0));
    return ((msum * u) * u) * u;
}
// ... and the step from the sum of the terms (added one by one, i outer over all bodies, j inner over the planets)
__device__ __forceinline__ double verlet_step_of(const double sum, const double timestep_scale, const double max_timestep) {
#pragma clang fp contract(off)
    return rsqrt_full(__builtin_fma(sum, timestep_scale * timestep_scale, 1.0)) * max_timestep;
}
// VerletPropagator::advance (verlet.hpp:139-175), one coordinate: half a drift with the old step, ...
__device__ __forceinline__ void verlet_first(const double h_first_half, const double v, double &p) {
#pragma clang fp contract(off)
    p = __builtin_fma(h_first_half, v, p);
}
// ... the kick in two halves (old and new step) and half a drift with the new one
__device__ __forceinline__ void verlet_second(const double h_first_half, const double h_second_half, const double acc,
                                              double &p, double &v) {
#pragma clang fp contract(off)
    double vel = __builtin_fma(h_first_half, acc, v);
    vel = __builtin_fma(h_second_half, acc, vel);
    v = vel;
    p = __builtin_fma(h_second_half, vel, p);
}

} // namespace swh
