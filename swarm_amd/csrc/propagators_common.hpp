// propagators_common.hpp -- the per-coordinate arithmetic of the tutorial propagators (propagators/euler.hpp:101-122,
// midpoint.hpp:106-160), shared by the lane-per-system kernel (kernels_generic.hip) and the lane-per-body kernel
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

} // namespace swh
