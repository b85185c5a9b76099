// kernels_rkck.hip -- fused Cash-Karp Runge-Kutta kernels (fixed / adaptive) for gfx950.
//
// Replaces rkck<Adapt,Monitor,Grav>::kernel (integrators/rkck_integrator.hpp:90-294).  Step
// control is per lane: every system carries its own time_step, accepts or rejects on its own and
// retires on its own; the wavefront runs until its last system is done or the iteration budget
// (destination_time / max_time_step attempts, rkck_integrator.hpp:75,143) is used up.
//
// Deviations from the reference, on purpose: (1) stage velocities are not stored, see the comment
// above rkck_kernel; (2) after a REJECTED attempt the reference leaves
// p6^2, v6^2 in the global ensemble (L221) until the next force evaluation overwrites them, so a
// launch whose budget ends on a reject hands back squares.  Here state is only written from the
// last ACCEPTED step.
#include "launch.hpp"

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
// The monitor proper, out of line.  It is called once per accepted step at most -- six force evaluations apart --
// and, for stop_on_ejection / log_time_interval, only when the wave-uniform early-out below fails; inlined, its
// code costs the attempt loop registers (C3: 9.9 ms with the monitor armed against 8.5 ms without).  The state
// travels through a stack copy so that `s` itself never has its address taken.
template <int N, bool VERBOSE>
__device__ __attribute__((noinline)) void rkck_mon_call(MonState *ms, State<N> *s, const MonParams *mp, const LogBuf *lg,
                                                        bool participate) {
    mon_call<N, VERBOSE>(*ms, *s, *mp, *lg, participate);
}

template <int N, bool ADAPTIVE, int MON>
__global__ __launch_bounds__(256) void rkck_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
                                                   Counters cnt) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    State<N> s;
    load_state(e, gid, s);
    const int state_in = s.state;
    MonState ms;
    mon_init(ms, s);
    if (MON) mon_call<N, MON == MON_VERBOSE>(ms, s, mp, lg, true); // rkck_integrator.hpp:140, every system

    const int iteration_count = (int)(ip.destination_time / ip.max_time_step); // L75
    const double dest = ip.destination_time;
    double time_step = ip.max_time_step; // L133
    unsigned long long accepted = 0, attempted = 0;

    // The wave stays whole: every lane goes through every attempt until the LAST system of the wave is done, and a
    // lane that is done (or was not ACTIVE at launch) simply never commits -- its registers keep the state that is
    // stored at the end.  Leaving the loop lane by lane would thin the wave out, and MI355X issues FP64 VALU
    // instructions at less than a third of the rate once 8 or fewer lanes of a wavefront are enabled (measured,
    // tools/ubench_exec.hip: 5.3 cycles per instruction with 9..64 lanes enabled, 16.9 with 1..8, whichever lanes they
    // are).  A tail of a few slow systems per wave -- the normal end of an adaptive run -- ran 4.4x slower per attempt.
    bool live = s.state == SWARM_HIP_ACTIVE;
    for (int iter = 0; iter < iteration_count && __any(live); iter++) {
        double h = time_step;
        {
#pragma clang fp contract(off)
            if (s.t + h > dest) h = dest - s.t;
        }
        // fixed step: every attempt is committed, so a lane that only keeps the wave whole takes h = 0, which leaves
        // its registers exactly as they are (p + 0 * (...) = p); the adaptive form withholds the commit instead
        if (!ADAPTIVE && !live) h = 0.0;
        double ka[6][N][3];
        double sp[N][3];
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 3; c++) sp[b][c] = s.p[b][c];

        // stages 1..6: ka[k] = acc(p_k); p_{k+1} from (pos, vel, ka[0..k])
#pragma unroll
        for (int k = 0; k < 6; k++) {
            acc_only<N, false>(sp, s.m, ka[k]);
            if (k < 5) {
                const double ck = CK::rowsum(CK::b[k + 1], 6);
#pragma unroll
                for (int b = 0; b < N; b++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        double acc2 = 0.0;
#pragma unroll
                        for (int j = 0; j < k; j++) { // second(b[k+1], j) is zero for j >= k
                            const double w = CK::second(CK::b[k + 1], 6, j);
                            acc2 = (j == 0) ? w * ka[j][b][c] : __builtin_fma(w, ka[j][b][c], acc2);
                        }
                        // pos + (c_k h) vel + h^2 sum: one operation per component fewer than h (c_k vel + h sum)
                        const double lin = __builtin_fma(ck * h, s.v[b][c], s.p[b][c]);
                        sp[b][c] = (k == 0) ? lin : __builtin_fma(h * h, acc2, lin);
                    }
            }
        }
        // 5th-order solution (p6, v6) and the embedded error estimate (L193-199)
        double p6[N][3], v6[N][3];
        double max_error = 0;
#pragma unroll
        for (int b = 0; b < N; b++) {
            double pem = 0, vem = 0, pm = 0, vm = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double sa = 0, sa2 = 0, ea = 0, ea2 = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const double kj = ka[j][b][c];
                    sa = (j == 0) ? CK::b6[0] * kj : __builtin_fma(CK::b6[j], kj, sa);
                    sa2 = (j == 0) ? CK::second(CK::b6, 6, 0) * kj : __builtin_fma(CK::second(CK::b6, 6, j), kj, sa2);
                    if (ADAPTIVE) {
                        ea = (j == 0) ? CK::ecc[0] * kj : __builtin_fma(CK::ecc[j], kj, ea);
                        ea2 = (j == 0) ? CK::second(CK::ecc, 6, 0) * kj : __builtin_fma(CK::second(CK::ecc, 6, j), kj, ea2);
                    }
                }
                v6[b][c] = __builtin_fma(h, sa, s.v[b][c]);
                p6[b][c] = __builtin_fma(h * h, sa2, __builtin_fma(CK::rowsum(CK::b6, 6) * h, s.v[b][c], s.p[b][c]));
                if (ADAPTIVE) {
                    const double ve = h * ea;
                    const double pe = __builtin_fma(h * h, ea2, (CK::rowsum(CK::ecc, 6) * h) * s.v[b][c]);
                    pem = (c == 0) ? pe * pe : pem + pe * pe;
                    vem = (c == 0) ? ve * ve : vem + ve * ve;
                    pm = (c == 0) ? p6[b][c] * p6[b][c] : pm + p6[b][c] * p6[b][c];
                    vm = (c == 0) ? v6[b][c] * v6[b][c] : vm + v6[b][c] * v6[b][c];
                }
            }
            if (ADAPTIVE) {
                // L232-238; quotients through the 1-ulp reciprocal path: the estimate itself is a difference of
                // nearly equal sums and is good to a few digits only
                const double pen = fdiv(pem, pm), ven = fdiv(vem, vm);
                max_error = fmax(fmax(pen, ven), max_error);
            }
        }
        bool accept_step = true;
        if (ADAPTIVE) { // L204-269: float constants, pow() on the normalised error
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
                const double step_change_factor =
                    ((normalized_error < 0.5) || (normalized_error > 1.0))
                        ? step_guess_safety_factor * pow(normalized_error, 0.5 * step_guess_power)
                        : 1.0;
                new_time_step =
                    (normalized_error > 1.)
                        ? fmax(time_step * fmax(step_change_factor, (double)step_shrink_min_factor),
                               ip.min_time_step)
                        : fmin(time_step * fmax(fmin(step_change_factor, (double)step_grow_max_factor), 1.0),
                               ip.max_time_step);
            }
            accept_step = (normalized_error < 1.0) || (fabs(time_step - new_time_step) < 1e-10);
            time_step = new_time_step;
        }
        if (ADAPTIVE) accept_step = accept_step && live;
        if (live) attempted++;
        if (accept_step) {
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) s.p[b][c] = p6[b][c], s.v[b][c] = v6[b][c];
            s.t += h;
            if (live) accepted++;
        }
        // monitor runs only for lanes that accepted (L283); the call itself is wave-wide
        if (MON != MON_OFF && MON != MON_ON) {
            mon_call<N, true>(ms, s, mp, lg, accept_step && live); // printing / every-step monitors: inline
        } else if (MON) {
            const bool stepped = accept_step && live;
            bool maybe = stepped;
            if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) {
#pragma clang fp contract(off)
                maybe = stepped && mp.log_interval != 0. && s.t >= ms.next_log_time;
            } else { // stop_on_ejection (the composite with a close-encounter part runs as MON_VERBOSE)
                maybe = false;
                if (stepped && mp.ej_any) {
#pragma clang fp contract(off)
#pragma unroll
                    for (int b = 1; b < N; b++) {
                        const double r2 = s.p[b][0] * s.p[b][0] + s.p[b][1] * s.p[b][1] + s.p[b][2] * s.p[b][2];
                        maybe = maybe || !(r2 < mp.r2_below);
                    }
                }
            }
            if (__any(maybe)) {
                State<N> tmp = s;
                MonState mtmp = ms;
                rkck_mon_call<N, false>(&mtmp, &tmp, &mp, &lg, stepped);
                s = tmp;
                ms = mtmp;
            }
        }
        if (accept_step && live && s.state == SWARM_HIP_ACTIVE && s.t >= dest) s.state = SWARM_HIP_INACTIVE;
        live = live && s.state == SWARM_HIP_ACTIVE;
    }
    if (accepted || s.state != state_in) store_state(e, gid, s);
    if (cnt.accepted) cnt.accepted[gid] += accepted;
    if (cnt.attempted) cnt.attempted[gid] += attempted;
}

template <int N, bool ADAPTIVE>
static int launch_n(const LaunchCtx &c) {
#define SWH_LAUNCH_VARIANT(V) \
    hipLaunchKernelGGL((rkck_kernel<N, ADAPTIVE, V>), dim3(c.grid), dim3(c.block), 0, c.stream, c.ens, c.ip, c.mp, c.lg, c.cnt)
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
    return check_launch("rkck kernel launch");
}

int launch_rkck(const LaunchCtx &c, bool adaptive) {
    if (adaptive) {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, true>(c)));
    } else {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, false>(c)));
    }
    return SWARM_HIP_OK;
}

} // namespace swh
