// mce_monitor.hpp -- the mce_stat monitor (Mercury close-encounter / collision statistics,
// monitors/mercury_mce_stat.hpp) for kernels that own one system per lane.  Used by `hermite`
// (kernels_mce.hip) and `irk2` (kernels_irk2.hip), the two integrators the reference binds it to
// (plugins/hermite.cu:40, plugins/irk2.cu:39).
//
// Per step, the lane forms the bounding boxes of its planets' paths (pass_one, L192-219) and the
// trigger conditions of pass_two on registers with compile-time loops; only when a box pair overlaps
// or a sun-planet condition holds does it copy the system into a small private work area and run
// pass_two (L222-433) there: the pair scan can be sent back to an earlier row by a merger (McePairScan),
// so it needs run-time body indices, which registers cannot give.
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

// ---- pass_two (monitors/mercury_mce_stat.hpp:222-433), as a scan over planet pairs and one over the planets --------
// The arithmetic keeps the reference's precision to the letter -- which intermediate is a float, which a double, no
// FMA contraction -- because a merge decision is a comparison and every printed line and merged mass is checked
// against the CPU restatement.  The organisation is this file's: what happens to a pair, what happens to a planet
// near the star, and the order in which pairs are visited are separate pieces.

// smallest squared separation of planets a, b during the step and the time (<= 0, relative to the step's end) at
// which it is reached: an end point when the pair recedes at the start and approaches at the end, else the minimum
// of the cubic Hermite interpolant of the squared separation, whose coefficients the reference holds in floats
template <int N>
__device__ __forceinline__ void mce_closest_approach(const MceWork<N> &w, const int a, const int b, const double h,
                                                     double &d2min, double &tmin) {
#pragma clang fp contract(off)
    double sep[2] = {0, 0}, rate[2] = {0, 0}; // |dr|^2 and d|dr|^2/dt at the start [0] and the end [1] of the step
    {
        const double rx = w.p0[a][0] - w.p0[b][0], ry = w.p0[a][1] - w.p0[b][1], rz = w.p0[a][2] - w.p0[b][2];
        const double ux = w.v0[a][0] - w.v0[b][0], uy = w.v0[a][1] - w.v0[b][1], uz = w.v0[a][2] - w.v0[b][2];
        rate[0] = (rx * ux + ry * uy + rz * uz) * 2.0;
        sep[0] = rx * rx + ry * ry + rz * rz;
    }
    {
        const double rx = w.p1[a][0] - w.p1[b][0], ry = w.p1[a][1] - w.p1[b][1], rz = w.p1[a][2] - w.p1[b][2];
        const double ux = w.v1[a][0] - w.v1[b][0], uy = w.v1[a][1] - w.v1[b][1], uz = w.v1[a][2] - w.v1[b][2];
        rate[1] = (rx * ux + ry * uy + rz * uz) * 2.0;
        sep[1] = rx * rx + ry * ry + rz * rz;
    }
    if (rate[0] > 0 && rate[1] < 0) {
        const bool at_start = sep[0] < sep[1];
        d2min = at_start ? sep[0] : sep[1];
        tmin = at_start ? -h : 0.0;
        return;
    }
    // stationary point of the interpolant: root of qa tau^2 + qb tau + qc, tau in [-1, 0] (L282-295)
    const float six_drop = (float)(6.0 * (sep[0] - sep[1]));
    const float qa = (float)(six_drop + 3.0 * h * (rate[0] + rate[1]));
    const float qb = (float)(six_drop + 2.0 * h * (rate[0] + 2.0 * rate[1]));
    const float qc = (float)(h * rate[1]);
    const float root_den = (float)(-0.5 * (qb + ((qb > 0) ? 1.0 : (-1.0)) * sqrt(fmax((double)(qb * qb - 4 * qa * qc), 0.0))));
    double tau = (root_den == 0.0) ? 0.0 : qc / root_den;
    tau = fmax(fmin(tau, 0.0), -1.0);
    tmin = tau * h;
    const float one_plus_tau = (float)(1.0 + tau);
    d2min = tau * tau * ((3.0 + 2.0 * tau) * sep[0] + one_plus_tau * h * rate[0]) +
            one_plus_tau * one_plus_tau * ((1.0 - 2.0 * tau) * sep[1] + tau * h * rate[1]);
    d2min = fmax(d2min, 0.0);
}

// inelastic merger of two planets at the end of the step: `keep` takes the total mass and the mass-weighted position
// and velocity (weights in float), `gone` is left massless, at rest, a hundred times farther out; the orbital energy
// of the pair goes to the system's attribute 0 (L316-336)
template <int N>
__device__ __forceinline__ void mce_merge_planets(MceWork<N> &w, const int keep, const int gone) {
#pragma clang fp contract(off)
    const double rx = w.p1[keep][0] - w.p1[gone][0], ry = w.p1[keep][1] - w.p1[gone][1], rz = w.p1[keep][2] - w.p1[gone][2];
    const double ux = w.v1[keep][0] - w.v1[gone][0], uy = w.v1[keep][1] - w.v1[gone][1], uz = w.v1[keep][2] - w.v1[gone][2];
    const float total = (float)(w.m[keep] + w.m[gone]);
    const float reduced = (float)(w.m[keep] * w.m[gone] / total);
    w.energy_lost = 0.5 * reduced * (ux * ux + uy * uy + uz * uz) - w.m[keep] * w.m[gone] / sqrt(rx * rx + ry * ry + rz * rz);
    w.energy_set = 1;
    const float share_keep = (float)(w.m[keep] / total), share_gone = (float)(w.m[gone] / total);
    for (int c = 0; c < 3; c++) {
        w.p1[keep][c] = w.p1[keep][c] * share_keep + w.p1[gone][c] * share_gone;
        w.v1[keep][c] = w.v1[keep][c] * share_keep + w.v1[gone][c] * share_gone;
        w.p1[gone][c] *= 100;
        w.v1[gone][c] = 0.0;
    }
    w.m[keep] = total;
    w.m[gone] = 0.0;
}

// a third body in a collision: report, snapshot, disable -- and go on scanning, as the reference does
template <int N>
__device__ __forceinline__ void mce_halt(MceWork<N> &w, const LogBuf &lg) {
    uint8_t *r = log_reserve_lane(lg, snapshot_len(N));
    if (r) snapshot_fill<N>(r, w.t, w.id, w.state, w.p1, w.v1, w.m);
    w.state = SWARM_HIP_DISABLED;
}

template <int N>
__device__ __forceinline__ bool mce_boxes_overlap(const MceWork<N> &w, const int a, const int b) {
    for (int c = 0; c < 3; c++)
        if (!(w.ne[a][c] >= w.sw[b][c] && w.ne[b][c] >= w.sw[a][c])) return false;
    return true;
}

// The order in which the reference visits planet pairs (lo, hi), 1 <= lo < hi < N: row by row, hi = 2 .. N-1, lo =
// 1 .. hi-1.  One irregularity is part of its observable behaviour: when the body with the LOWER index is the lighter
// one of a merging pair, the reference exchanges its two loop variables to name the survivor first (L312) and never
// exchanges them back, so its loops resume from row lo + 1: every pair of the rows lo+1 .. hi is visited again, with
// the boxes of the step's start and the absorbed body now massless -- which is how such a merger ends in "3 planets
// in collision: System halted" whenever the absorbed body's box still overlaps a neighbour's.  `rewind_after` is
// that irregularity, spelled out.
struct McePairScan {
    int lo, hi;
    __device__ McePairScan() : lo(1), hi(2) {}
    __device__ bool done(const int n) const { return hi >= n; }
    __device__ void next() {
        if (++lo >= hi) lo = 1, ++hi;
    }
    // the next() that follows lands on the first pair of row absorbed_lower_index + 1
    __device__ void rewind_after(const int absorbed_lower_index) { lo = hi = absorbed_lower_index; }
};

// planet j against the star over the step (L352-429): a pericentre passage (r.v changes sign from inward to outward)
// or a start/end point inside the star's radius; then, from the osculating orbit at the step's start in float
// arithmetic, accretion when the pericentre distance q = p / (1 + e) is inside the star
template <int N>
__device__ __forceinline__ void mce_planet_and_star(MceWork<N> &w, const int j, const double h, const LogBuf &lg) {
#pragma clang fp contract(off)
    const int star = 0;
    const double(&x0)[3] = w.p0[j], (&u0)[3] = w.v0[j], (&x1)[3] = w.p1[j], (&u1)[3] = w.v1[j];
    const float r2_start = (float)(x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2]);
    const float r2_end = (float)(x1[0] * x1[0] + x1[1] * x1[1] + x1[2] * x1[2]);
    const float rdotv_start = (float)(u0[0] * x0[0] + u0[1] * x0[1] + u0[2] * x0[2]);
    const float rdotv_end = (float)(u1[0] * x1[0] + u1[1] * x1[1] + x1[2] * x1[2]); // L361: z term is pos*pos there
    const bool pericentre_passed = rdotv_start * h <= -1e-6 && rdotv_end * h >= 1e-6;
    if (!pericentre_passed && !(fminf(r2_start, r2_end) <= w.radius0 * w.radius0)) return;
    if (!(w.m[j] > 0)) { // an absorbed body falling in as well
        const PArg none[1] = {parg(0)};
        const char f[] = "2 planets and the sun in collision: System halted !\n";
        log_printf_lane(lg, f, (int)sizeof(f), none, 0);
        mce_halt(w, lg);
        return;
    }
    const float hx = (float)(x0[1] * u0[2] - x0[2] * u0[1]);
    const float hy = (float)(x0[2] * u0[0] - x0[0] * u0[2]);
    const float hz = (float)(x0[0] * u0[1] - x0[1] * u0[0]);
    const float speed2 = (float)(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
    const float h2 = hx * hx + hy * hy + hz * hz;
    const double gm = w.m[star] + w.m[j];
    const float semilatus = (float)(h2 / gm);
    const float r_start = sqrtf(r2_start);
    const float ecc2 = (float)(1.0 + semilatus * (speed2 / gm - 2.0 / r_start));
    const float ecc = (float)sqrt(fmax((double)ecc2, 0.0));
    const float pericentre = (float)(semilatus / (1.0 + ecc));
    if (!(pericentre <= w.radius0)) return;
    SWH_LPRINTF(lg, "Hitting with the sun detected: sysid = %d, time = %f, j=%d\n", parg(w.id), parg(w.t), parg(j));
    // the star takes the planet's momentum share and mass; the planet is parked like an absorbed one; the other
    // planets are re-centred on the displaced star (L405-423)
    const float planet_share = (float)(w.m[j] / (w.m[j] + w.m[star]));
    for (int c = 0; c < 3; c++) {
        w.p1[star][c] = planet_share * x1[c];
        w.v1[star][c] = planet_share * u1[c];
    }
    w.m[star] = w.m[j] + w.m[star];
    for (int c = 0; c < 3; c++) {
        w.p1[j][c] *= 100.0;
        w.v1[j][c] *= 0.0;
    }
    w.m[j] = 0.0;
    for (int k = 1; k < N; k++) {
        if (k == j) continue;
        for (int c = 0; c < 3; c++) {
            w.p1[k][c] = w.p1[k][c] - w.p1[star][c];
            w.v1[k][c] = w.v1[k][c] - w.v1[star][c];
        }
    }
}

template <int N>
__device__ __noinline__ void mce_pass_two(MceWork<N> &w, const double timestep, const LogBuf lg) {
    for (McePairScan scan; !scan.done(N); scan.next()) {
        const int lo = scan.lo, hi = scan.hi;
        if (!mce_boxes_overlap(w, lo, hi)) continue;
        if (!(w.m[lo] > 0 && w.m[hi] > 0)) { // one of the two was absorbed earlier in this pass
            SWH_LPRINTF(lg, "3 planets in collision: System halted ! T=%f\n", parg(w.t));
            mce_halt(w, lg);
            continue;
        }
        double d2min, tmin;
        mce_closest_approach(w, lo, hi, timestep, d2min, tmin);
        float reach = (float)fmax(w.rphys[lo], w.rphys[hi]);
        reach = reach * reach;
        if (!(d2min <= reach)) continue;
        const bool lower_is_lighter = w.m[lo] < w.m[hi];
        const int keep = lower_is_lighter ? hi : lo, gone = lower_is_lighter ? lo : hi;
        SWH_LPRINTF(lg, "Hitting detected: sysid = %d, i = %d, j = %d, time= %f\n", parg(w.id), parg(keep), parg(gone),
                    parg(w.t + tmin));
        mce_merge_planets(w, keep, gone);
        if (lower_is_lighter) scan.rewind_after(lo);
    }
    for (int j = 1; j < N; j++) mce_planet_and_star(w, j, timestep, lg);
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
