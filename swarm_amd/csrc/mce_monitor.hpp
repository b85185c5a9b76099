// mce_monitor.hpp -- the mce_stat monitor (Mercury close-encounter / collision statistics,
// monitors/mercury_mce_stat.hpp) for kernels that own one system per lane.  Used by `hermite`
// (kernels_mce.hip) and `irk2` (kernels_irk2.hip), the two integrators the reference binds it to
// (plugins/hermite.cu:40, plugins/irk2.cu:39).
//
// Per step, the lane forms the bounding boxes of its planets' paths (pass_one, L192-219) and the
// trigger conditions of pass_two on registers with compile-time loops; only when a box pair overlaps
// or a sun-planet condition holds does it copy the system into a small private work area and run the
// reference's pass_two (L222-433) literally -- that code swaps its own loop variables (L312), so it
// needs run-time body indices, which registers cannot give.
#pragma once
#include "launch.hpp"


namespace swh {

#define MCE_PI 3.14159265358979      // mercury_mce_stat.hpp:26
#define MCE_THIRD 0.3333333333333333 // L27

template <int N>
struct MceWork {
    double p0[N][3], v0[N][3]; // start of the step (the monitor's shared[b][c].pos/vel)
    double p1[N][3], v1[N][3]; // end of the step   (the global system, rewritten by merges)
    double m[N];
    double sw[N][3], ne[N][3]; // bounding boxes
    double rphys[N];           // attribute(2)
    double radius0;            // _sys[0].attribute(0)
    double t;
    double energy_lost;        // _sys.attribute(0)
    int id, state;
    int energy_set;
};

// mce_stat::pass_two, L222-433, thread 0's serial code.  Mixed float/double arithmetic as written
// in the reference; no FMA contraction.
template <int N>
__device__ __noinline__ void mce_pass_two(MceWork<N> &w, const double timestep, const LogBuf lg) {
#pragma clang fp contract(off)
    int i, j;
    // planet - planet
    for (j = 2; j < N; j++)
        for (i = 1; i < j; i++) {
            if (w.ne[i][0] >= w.sw[j][0] && w.ne[j][0] >= w.sw[i][0])
                if (w.ne[i][1] >= w.sw[j][1] && w.ne[j][1] >= w.sw[i][1])
                    if (w.ne[i][2] >= w.sw[j][2] && w.ne[j][2] >= w.sw[i][2]) {
                        if (w.m[i] > 0 && w.m[j] > 0) {
                            const double dx0 = w.p0[i][0] - w.p0[j][0], dy0 = w.p0[i][1] - w.p0[j][1],
                                         dz0 = w.p0[i][2] - w.p0[j][2];
                            const double du0 = w.v0[i][0] - w.v0[j][0], dv0 = w.v0[i][1] - w.v0[j][1],
                                         dw0 = w.v0[i][2] - w.v0[j][2];
                            const double d0t = (dx0 * du0 + dy0 * dv0 + dz0 * dw0) * 2.0;
                            const double dx1 = w.p1[i][0] - w.p1[j][0], dy1 = w.p1[i][1] - w.p1[j][1],
                                         dz1 = w.p1[i][2] - w.p1[j][2];
                            const double du1 = w.v1[i][0] - w.v1[j][0], dv1 = w.v1[i][1] - w.v1[j][1],
                                         dw1 = w.v1[i][2] - w.v1[j][2];
                            const double d1t = (dx1 * du1 + dy1 * dv1 + dz1 * dw1) * 2.0;
                            const double d0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;
                            const double d1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
                            double d2min, tmin, tau;
                            if (d0t > 0 && d1t < 0) {
                                if (d0 < d1) {
                                    d2min = d0;
                                    tmin = -timestep;
                                } else {
                                    d2min = d1;
                                    tmin = 0.0;
                                }
                            } else { // Hermite interpolation of the separation; float coefficients (L282-285)
                                float tmp = (float)(6.0 * (d0 - d1));
                                const float a = (float)(tmp + 3.0 * timestep * (d0t + d1t));
                                const float b = (float)(tmp + 2.0 * timestep * (d0t + 2.0 * d1t));
                                const float c = (float)(timestep * d1t);
                                tmp = (float)(-0.5 * (b + ((b > 0) ? 1.0 : (-1.0)) * sqrt(fmax((double)(b * b - 4 * a * c), 0.0))));
                                if (tmp == 0.0)
                                    tau = 0.0;
                                else
                                    tau = c / tmp;
                                tau = fmin(tau, 0.0);
                                tau = fmax(tau, -1.0);
                                tmin = tau * timestep;
                                tmp = (float)(1.0 + tau);
                                d2min = tau * tau * ((3.0 + 2.0 * tau) * d0 + tmp * timestep * d0t) +
                                        tmp * tmp * ((1.0 - 2.0 * tau) * d1 + tau * timestep * d1t);
                                d2min = fmax(d2min, 0.0);
                            }
                            float d2hit = (float)fmax(w.rphys[i], w.rphys[j]);
                            d2hit = d2hit * d2hit;
                            if (d2min <= d2hit) {
                                if (w.m[i] < w.m[j]) { // L312 swaps the loop variables themselves
                                    const int k = i;
                                    i = j;
                                    j = k;
                                }
                                SWH_LPRINTF(lg, "Hitting detected: sysid = %d, i = %d, j = %d, time= %f\n", parg(w.id),
                                            parg(i), parg(j), parg(w.t + tmin));
                                const float msum = (float)(w.m[i] + w.m[j]);
                                const float mredu = (float)(w.m[i] * w.m[j] / msum);
                                w.energy_lost = 0.5 * mredu * (du1 * du1 + dv1 * dv1 + dw1 * dw1) -
                                                w.m[i] * w.m[j] / sqrt(dx1 * dx1 + dy1 * dy1 + dz1 * dz1);
                                w.energy_set = 1;
                                const float tmp1 = (float)(w.m[i] / msum);
                                const float tmp2 = (float)(w.m[j] / msum);
                                for (int c = 0; c < 3; c++) {
                                    w.p1[i][c] = w.p1[i][c] * tmp1 + w.p1[j][c] * tmp2;
                                    w.v1[i][c] = w.v1[i][c] * tmp1 + w.v1[j][c] * tmp2;
                                }
                                w.m[i] = msum;
                                for (int c = 0; c < 3; c++) {
                                    w.p1[j][c] *= 100;
                                    w.v1[j][c] = 0.0;
                                }
                                w.m[j] = 0.0;
                            }
                        } else {
                            SWH_LPRINTF(lg, "3 planets in collision: System halted ! T=%f\n", parg(w.t));
                            uint8_t *r = log_reserve_lane(lg, snapshot_len(N));
                            if (r) snapshot_fill<N>(r, w.t, w.id, w.state, w.p1, w.v1, w.m);
                            w.state = SWARM_HIP_DISABLED;
                        }
                    }
        }
    // sun - planet
    i = 0;
    for (j = 1; j < N; j++) {
        const float rr0 = (float)(w.p0[j][0] * w.p0[j][0] + w.p0[j][1] * w.p0[j][1] + w.p0[j][2] * w.p0[j][2]);
        const float rr1 = (float)(w.p1[j][0] * w.p1[j][0] + w.p1[j][1] * w.p1[j][1] + w.p1[j][2] * w.p1[j][2]);
        const float rv0 = (float)(w.v0[j][0] * w.p0[j][0] + w.v0[j][1] * w.p0[j][1] + w.v0[j][2] * w.p0[j][2]);
        // L361: the z term is pos*pos in the reference
        const float rv1 = (float)(w.v1[j][0] * w.p1[j][0] + w.v1[j][1] * w.p1[j][1] + w.p1[j][2] * w.p1[j][2]);
        if ((rv0 * timestep <= -1e-6 && rv1 * timestep >= 1e-6) || (fminf(rr0, rr1) <= w.radius0 * w.radius0)) {
            if (w.m[j] > 0) {
                const float hx = (float)(w.p0[j][1] * w.v0[j][2] - w.p0[j][2] * w.v0[j][1]);
                const float hy = (float)(w.p0[j][2] * w.v0[j][0] - w.p0[j][0] * w.v0[j][2]);
                const float hz = (float)(w.p0[j][0] * w.v0[j][1] - w.p0[j][1] * w.v0[j][0]);
                const float v2 = (float)(w.v0[j][0] * w.v0[j][0] + w.v0[j][1] * w.v0[j][1] + w.v0[j][2] * w.v0[j][2]);
                const float h2 = hx * hx + hy * hy + hz * hz;
                const float p = (float)(h2 / (w.m[i] + w.m[j]));
                const float r0 = sqrtf(rr0);
                const float temp = (float)(1.0 + p * (v2 / (w.m[i] + w.m[j]) - 2.0 / r0));
                const float e = (float)sqrt(fmax((double)temp, 0.0));
                const float q = (float)(p / (1.0 + e));
                if (q <= w.radius0) {
                    SWH_LPRINTF(lg, "Hitting with the sun detected: sysid = %d, time = %f, j=%d\n", parg(w.id), parg(w.t),
                                parg(j));
                    const float tmp2 = (float)(w.m[j] / (w.m[j] + w.m[i]));
                    for (int c = 0; c < 3; c++) {
                        w.p1[i][c] = tmp2 * w.p1[j][c];
                        w.v1[i][c] = tmp2 * w.v1[j][c];
                    }
                    w.m[i] = w.m[j] + w.m[i];
                    for (int c = 0; c < 3; c++) {
                        w.p1[j][c] *= 100.0;
                        w.v1[j][c] *= 0.0;
                    }
                    w.m[j] = 0.0;
                    for (int k = 1; k < N; k++)
                        if (k != j)
                            for (int c = 0; c < 3; c++) {
                                w.p1[k][c] = w.p1[k][c] - w.p1[0][c];
                                w.v1[k][c] = w.v1[k][c] - w.v1[0][c];
                            }
                }
            } else {
                const PArg none[1] = {parg(0)};
                const char f[] = "2 planets and the sun in collision: System halted !\n";
                log_printf_lane(lg, f, (int)sizeof(f), none, 0);
                uint8_t *r = log_reserve_lane(lg, snapshot_len(N));
                if (r) snapshot_fill<N>(r, w.t, w.id, w.state, w.p1, w.v1, w.m);
                w.state = SWARM_HIP_DISABLED;
            }
        }
    }
}

// rce (attribute 1), rphysics (attribute 2) of every body and the star's radius (attribute 0 of body 0)
template <int N>
struct MceData {
    double rce[N], rphys[N], radius0;
};

// mce_stat::init, L134-166: Hill radius -> rce and rphysics of every planet, written to the body
// attributes, for every system of the launch.  `0 < t < nbod` (L136) is always true, so thread 0 runs
// the planet block too and then overwrites its result with 0: body 0 gets zeros.
template <int N>
__device__ __forceinline__ void mce_init(const EnsView &e, int gid, const State<N> &s, const MonParams &mp,
                                         MceData<N> &md) {
    const int chunk = gid / W, lane = gid % W;
    double *const brow = e.body + (size_t)chunk * N * e.nbf * W + lane; // body b, field f: brow[(b*nbf+f)*W]
    md.radius0 = brow[(size_t)7 * W];
    md.rce[0] = md.rphys[0] = 0.0;
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int b = 1; b < N; b++) {
            const double gm = s.m[0] + s.m[b];
            const double r = sqrt(s.p[b][0] * s.p[b][0] + s.p[b][1] * s.p[b][1] + s.p[b][2] * s.p[b][2]);
            const double vi_max = s.v[b][0] * s.v[b][0] + s.v[b][1] * s.v[b][1] + s.v[b][2] * s.v[b][2];
            double a = gm * r / (2 * gm - r * vi_max);
            if (a <= 0) a = r;
            const double hill = a * pow(MCE_THIRD * s.m[b] / s.m[0], MCE_THIRD);
            const double rad = brow[((size_t)b * e.nbf + 7) * W];
            const double rho = s.m[b] / (4 * MCE_THIRD * MCE_PI * rad * rad * rad);
            md.rce[b] = hill * mp.rceh;
            md.rphys[b] = hill / a * pow(2.25 * s.m[0] / (MCE_PI * rho), MCE_THIRD);
        }
    }
#pragma unroll
    for (int b = 0; b < N; b++) {
        brow[((size_t)b * e.nbf + 8) * W] = md.rce[b];
        brow[((size_t)b * e.nbf + 9) * W] = md.rphys[b];
    }
}

// mce_stat::operator()(thread, b, c, pos, vel), L168-186: (cp, cv) = the system at the START of the
// step, s = the system at its end.  Returns true when pass_two rewrote the system (merge / halt).
template <int N>
__device__ __forceinline__ bool mce_monitor(const EnsView &e, int gid, State<N> &s, const double (&cp)[N][3],
                                            const double (&cv)[N][3], const MceData<N> &md, const double timestep,
                                            const LogBuf &lg) {
    double sw[N][3], ne[N][3];
    bool trigger = false;
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int b = 1; b < N; b++) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double sp = cp[b][c], sv = cv[b][c], gp = s.p[b][c], gv = s.v[b][c];
                double lo = fmin(sp, gp), hi = fmax(sp, gp);
                if ((sv < 0 && gv > 0) || (sv > 0 && gv < 0)) {
                    const double temp = sv * gp - gv * sp - 0.5 * timestep * sv * gv / (sv - gv);
                    lo = fmin(lo, temp);
                    hi = fmax(hi, temp);
                }
                const double tmp = md.rce[b] * 1.2;
                sw[b][c] = lo - tmp;
                ne[b][c] = hi + tmp;
            }
        }
#pragma unroll
        for (int j = 2; j < N; j++)
#pragma unroll
            for (int i = 1; i < j; i++)
                trigger = trigger || (ne[i][0] >= sw[j][0] && ne[j][0] >= sw[i][0] && ne[i][1] >= sw[j][1] &&
                                      ne[j][1] >= sw[i][1] && ne[i][2] >= sw[j][2] && ne[j][2] >= sw[i][2]);
#pragma unroll
        for (int j = 1; j < N; j++) {
            const float rr0 = (float)(cp[j][0] * cp[j][0] + cp[j][1] * cp[j][1] + cp[j][2] * cp[j][2]);
            const float rr1 = (float)(s.p[j][0] * s.p[j][0] + s.p[j][1] * s.p[j][1] + s.p[j][2] * s.p[j][2]);
            const float rv0 = (float)(cv[j][0] * cp[j][0] + cv[j][1] * cp[j][1] + cv[j][2] * cp[j][2]);
            const float rv1 = (float)(s.v[j][0] * s.p[j][0] + s.v[j][1] * s.p[j][1] + s.p[j][2] * s.p[j][2]);
            trigger = trigger || (rv0 * timestep <= -1e-6 && rv1 * timestep >= 1e-6) ||
                      (fminf(rr0, rr1) <= md.radius0 * md.radius0);
        }
    }
    if (!trigger) return false;
    MceWork<N> w;
    for (int b = 0; b < N; b++) {
        for (int c = 0; c < 3; c++) {
            w.p0[b][c] = cp[b][c], w.v0[b][c] = cv[b][c];
            w.p1[b][c] = s.p[b][c], w.v1[b][c] = s.v[b][c];
            w.sw[b][c] = b ? sw[b][c] : 0.0, w.ne[b][c] = b ? ne[b][c] : 0.0;
        }
        w.m[b] = s.m[b];
        w.rphys[b] = md.rphys[b];
    }
    w.radius0 = md.radius0;
    w.t = s.t, w.id = s.id, w.state = s.state;
    w.energy_lost = 0.0, w.energy_set = 0;
    mce_pass_two<N>(w, timestep, lg);
    for (int b = 0; b < N; b++) {
        for (int c = 0; c < 3; c++) s.p[b][c] = w.p1[b][c], s.v[b][c] = w.v1[b][c];
        s.m[b] = w.m[b];
    }
    s.state = w.state;
    if (w.energy_set) e.sysd[(size_t)(gid / W) * e.nsf * W + (gid % W) + (size_t)W] = w.energy_lost;
    return true;
}

// store_state() does not write masses (no other monitor changes them)
template <int N>
__device__ __forceinline__ void mce_store_masses(const EnsView &e, int gid, const State<N> &s) {
    double *const brow = e.body + (size_t)(gid / W) * N * e.nbf * W + (gid % W);
#pragma unroll
    for (int b = 0; b < N; b++) brow[((size_t)b * e.nbf + 6) * W] = s.m[b];
}

inline int mce_check_attributes(const LaunchCtx &c) {
    if (c.ens.nbf < 7 + 3 || c.ens.nsf < 1 + 1) {
        set_error("mce_stat needs 3 body attributes (radius, rce, rphysics) and 1 system attribute; "
                  "got nbattr=%d nsattr=%d (monitors/mercury_mce_stat.hpp:153-156,319)",
                  c.ens.nbf - 7, c.ens.nsf - 1);
        return SWARM_HIP_EINVAL;
    }
    return SWARM_HIP_OK;
}

} // namespace swh
