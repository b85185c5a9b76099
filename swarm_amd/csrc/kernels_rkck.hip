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
#include <climits>

#include "body_monitor.hpp"
#include "launch.hpp"
#include "rkck_common.hpp"

namespace swh {

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
        // (a lane that only keeps the wave whole computes its attempt like the others and never commits it)
        double ka[6][N][3];
        double sp[N][3];
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 3; c++) sp[b][c] = s.p[b][c];

        // stages 1..6: ka[k] = acc(p_k); p_{k+1} from (pos, vel, ka[0..k-1]) -- the operations of rkck_stage_position<k>
        // (rkck_common.hpp) in place: through the helper the same instructions schedule 1.3 % slower here (C3, A/B)
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
            double pe[3], ve[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                // (the operations of rkck_solution<ADAPTIVE>, rkck_common.hpp, in place)
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
                    ve[c] = h * ea;
                    pe[c] = __builtin_fma(h * h, ea2, (CK::rowsum(CK::ecc, 6) * h) * s.v[b][c]);
                }
            }
            if (ADAPTIVE) max_error = fmax(rkck_body_error(pe, ve, p6[b], v6[b]), max_error); // L232-238
        }
        bool accept_step = true;
        if (ADAPTIVE) accept_step = rkck_step_control(max_error, time_step, ip); // L204-269
        accept_step = accept_step && live;
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
                        maybe = maybe || ej_maybe_body(s.p[b], s.v[b], mp);
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

// One lane per body (kernels_rkck_body.hip) pays while the ensemble is too small to fill the chip with wavefronts of
// whole systems, and at nbod 9 and 10 -- where the six stage accelerations of all bodies spill to scratch with one lane
// per system -- at every size (65536 systems: 156 -> 57 ms and 196 -> 68 ms per 200 attempts).  Crossovers measured
// with WORKLOAD=c3 tools/lane_sweep.sh (profiles/r2/rkck_lane_sweep.txt); at 2048 systems the lane-per-body kernel
// is 1.2x (nbod 3) to 15x (nbod 10) faster.
static bool small_rkck_ensemble(const LaunchCtx &c) {
    static const int limit[SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM + 1] = {0, 0, 0, 16384, 32768, 24576, 20480, 40960, 49152, INT_MAX, INT_MAX};
    return within_limit(c, limit);
}

int launch_rkck(const LaunchCtx &c, bool adaptive) {
    // small ensembles with nothing to monitor, or stop_on_ejection without printing
    const int layout = choose_layout(c, body_monitor_of(c.mp, c.ens.nbod) != MONB_NONE, small_rkck_ensemble(c));
    if (layout < 0) return layout;
    if (layout == 1) return launch_rkck_body(c, adaptive);
    if (adaptive) {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, true>(c)));
    } else {
        SWH_DISPATCH_NBOD(c.ens.nbod, return (launch_n<N, false>(c)));
    }
    return SWARM_HIP_OK;
}

} // namespace swh
