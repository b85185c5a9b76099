// device_common.hpp -- shared device code of libswarm_hip.so (gfx950 only).
//
// Design: ONE LANE OWNS ONE SYSTEM.  A wavefront integrates the 64 systems of one "wave chunk";
// the whole state of a system (positions, velocities, accelerations, jerks, masses, time, state)
// lives in that lane's VGPRs for every step of the launch, so the reference's per-step global
// round trips, shared-memory pair tables and __syncthreads (swarm/gpu/gravitation_accjerk.hpp:186-197,
// SURVEY 3.3) disappear.  HBM is touched once per launch: coalesced 512-byte rows per field.
// (Small ensembles of hermite / hermite_adap: one lane per BODY instead, kernels_hermite_body.hip, same bits.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "swarm_hip.h"

namespace swh {

constexpr int W = SWARM_HIP_CHUNK;

struct EnsView {
    int nsys, nbod;
    int nbf; // doubles per body row group: 7 + nbattr
    int nsf; // 1 + nsattr
    double *body;
    double *sysd;
    int32_t *sysi;
    // Active list (lane-per-body kernels only, NULL otherwise): the systems that were ACTIVE when the launch was
    // prepared, gathered into alist[0 .. *alist_n) by active_list_kernel (capi.hip), so that a run whose ensemble has
    // thinned out -- the tail of an adaptive or monitored run -- works on dense wavefronts.
    const int32_t *alist;
    const int32_t *alist_n; // device count of ACTIVE systems found; at most alist_cap of them are in the list
    int alist_cap;
};

// Lane-per-body kernels: the system that slot `slot` of the launch works on (slot = workgroup * systems-per-wave +
// group of the lane).  Without a list slot k is system k.  `lane_ok` = the lane belongs to a whole group of its wave.
__device__ __forceinline__ bool body_slot_system(const EnsView &e, const bool lane_ok, const long long slot, int &sys) {
    bool ok;
    if (e.alist) {
        const int n = *e.alist_n;
        ok = lane_ok && slot < (long long)(n < e.alist_cap ? n : e.alist_cap);
        sys = ok ? e.alist[slot] : 0;
    } else {
        ok = lane_ok && slot < (long long)e.nsys;
        sys = ok ? (int)slot : 0;
    }
    return ok;
}

struct MonParams {
    int kind;
    int ej_any, ej_deactivate, ej_log;
    int ce_any, ce_deactivate, ce_log;
    double rmax;
    double r2_below; // r2 < r2_below  =>  sqrt(r2) < rmax for certain (host-computed lower bound)
    double dmin3;    // close_approach^3
    double log_interval;
    int ej_verbose, ce_verbose; // verbose_on_*: lprintf records (gpulog MSG_PRINTF)
    double dmin;                // close_approach
    double rceh, mce_time_step; // mce_stat_params (monitors/mercury_mce_stat.hpp:36-55)
    // log_transit_params (monitors/log_transit.hpp:42-58); dt = the time_step= key
    double tr_tol, tr_dt;
    int tr_log_transit, tr_log_occultation;
};

struct LogBuf {
    uint8_t *buf;
    int32_t *cursor; // [0] bytes used, [1] dropped records, [2] lprintf records written, [3] reserved
    int32_t cap;
};

struct Counters {
    unsigned long long *accepted;
    unsigned long long *attempted;
};

struct IntParams {
    int max_iterations;
    double destination_time;
    double time_step, time_step_factor, min_time_step, max_time_step, error_tolerance;
    double max_timestep, timestep_scale;
    int kepler_compensated; // swarm_hip_launch_t::kepler_compensated (mvs)
};

// compile-time loop: f(IntC<0>{}), ..., f(IntC<N-1>{})
template <int I>
struct IntC {
    static constexpr int value = I;
};
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<N, I + 1>(f);
    }
}

// ---------------------------------------------------------------------------------------------
template <int N>
struct State {
    double p[N][3], v[N][3], m[N];
    double t;
    int state, id;
    int number; // index of the system in the ensemble (SystemRef::number())
};

template <int N>
__device__ __forceinline__ void load_state(const EnsView &e, int gid, State<N> &s) {
    const int chunk = gid / W, lane = gid % W;
    const double *b = e.body + (size_t)chunk * N * e.nbf * W + lane;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double *r = b + (size_t)i * e.nbf * W;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            s.p[i][c] = r[c * W];
            s.v[i][c] = r[(3 + c) * W];
        }
        s.m[i] = r[6 * W];
    }
    s.t = e.sysd[(size_t)chunk * e.nsf * W + lane];
    s.state = e.sysi[(size_t)chunk * 2 * W + lane];
    s.id = e.sysi[(size_t)chunk * 2 * W + W + lane];
    s.number = gid;
}

template <int N>
__device__ __forceinline__ void store_state(const EnsView &e, int gid, const State<N> &s) {
    const int chunk = gid / W, lane = gid % W;
    double *b = e.body + (size_t)chunk * N * e.nbf * W + lane;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double *r = b + (size_t)i * e.nbf * W;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            r[c * W] = s.p[i][c];
            r[(3 + c) * W] = s.v[i][c];
        }
    }
    e.sysd[(size_t)chunk * e.nsf * W + lane] = s.t;
    e.sysi[(size_t)chunk * 2 * W + lane] = s.state;
}

// ---------------------------------------------------------------------------------------------
// 1/sqrt(x) to double precision: hardware v_rsq_f64 seed + one third-order Newton step.
// Replaces CUDA rsqrt() in `rsqrt(r2)/r2` (swarm/gpu/gravitation_accjerk.hpp:106).
__device__ __forceinline__ double rsqrt_full(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double t = x * y;
    double e = __builtin_fma(-t, y, 1.0);       // 1 - x*y^2
    double q = __builtin_fma(0.375, e, 0.5);    // 1/2 + 3/8 e
    return __builtin_fma(y * e, q, y);          // y*(1 + e/2 + 3/8 e^2)
}

// 1/x and a/b, sqrt(x) to ~1 ulp from the hardware seeds (v_rcp_f64 / v_rsq_f64, 2^-24) and one
// third-order Newton step.  Measured on MI355X (profiles/r1/fp64_valu_ubench.txt): the IEEE
// divide costs ~60 and the IEEE sqrt ~85 cycles per wave-instruction; these cost ~35 / ~45.
// Used where the reference's CUDA code has plain `/` and sqrt() in the MVS Kepler solver.
__device__ __forceinline__ double rcp_full(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    double t = __builtin_fma(e, e, e); // e + e^2
    return __builtin_fma(r, t, r);
}
__device__ __forceinline__ double fdiv(double a, double b) { return a * rcp_full(b); }
__device__ __forceinline__ double fsqrt(double x) { return x > 0.0 ? x * rsqrt_full(x) : (x == 0.0 ? 0.0 : __builtin_nan("")); }

// Pairwise acceleration + jerk, all bodies of one system in registers.
// Arithmetic follows hermite_cpu::calcForces (integrators/hermite_cpu.hpp:82-110): pairs i<j in
// lexicographic order, one accumulator per body, scalar = r^-3 * m_other; with
// r^-3 = rsqrt(r2)^3 and rv = 3 (dx.dv) rsqrt(r2)^2 instead of the reference's two divisions.
template <int N>
__device__ __forceinline__ void accjerk(const double (&p)[N][3], const double (&v)[N][3],
                                        const double (&m)[N], double (&a)[N][3], double (&jk)[N][3]) {
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
#pragma unroll
        for (int j = i + 1; j < N; j++) {
            const double dx0 = p[j][0] - p[i][0], dx1 = p[j][1] - p[i][1], dx2 = p[j][2] - p[i][2];
            const double dv0 = v[j][0] - v[i][0], dv1 = v[j][1] - v[i][1], dv2 = v[j][2] - v[i][2];
            const double r2 = __builtin_fma(dx2, dx2, __builtin_fma(dx1, dx1, dx0 * dx0));
            const double xv = __builtin_fma(dx2, dv2, __builtin_fma(dx1, dv1, dx0 * dv0));
            const double y = rsqrt_full(r2);
            const double y2 = y * y;
            const double rinv = y * y2;           // r^-3
            const double rv = (xv * 3.0) * y2;    // 3 (dx.dv) / r2
            const double si = rinv * m[j];
            const double sj = -(rinv * m[i]);
            const double t0 = __builtin_fma(-dx0, rv, dv0), t1 = __builtin_fma(-dx1, rv, dv1),
                         t2 = __builtin_fma(-dx2, rv, dv2);
            const bool first_i = (i == 0 && j == 1); // compile-time after unrolling
            const bool first_j = (i == 0);
            if (first_i) {
                a[i][0] = dx0 * si, a[i][1] = dx1 * si, a[i][2] = dx2 * si;
                jk[i][0] = t0 * si, jk[i][1] = t1 * si, jk[i][2] = t2 * si;
            } else {
                a[i][0] = __builtin_fma(dx0, si, a[i][0]);
                a[i][1] = __builtin_fma(dx1, si, a[i][1]);
                a[i][2] = __builtin_fma(dx2, si, a[i][2]);
                jk[i][0] = __builtin_fma(t0, si, jk[i][0]);
                jk[i][1] = __builtin_fma(t1, si, jk[i][1]);
                jk[i][2] = __builtin_fma(t2, si, jk[i][2]);
            }
            if (first_j) {
                a[j][0] = dx0 * sj, a[j][1] = dx1 * sj, a[j][2] = dx2 * sj;
                jk[j][0] = t0 * sj, jk[j][1] = t1 * sj, jk[j][2] = t2 * sj;
            } else {
                a[j][0] = __builtin_fma(dx0, sj, a[j][0]);
                a[j][1] = __builtin_fma(dx1, sj, a[j][1]);
                a[j][2] = __builtin_fma(dx2, sj, a[j][2]);
                jk[j][0] = __builtin_fma(t0, sj, jk[j][0]);
                jk[j][1] = __builtin_fma(t1, sj, jk[j][1]);
                jk[j][2] = __builtin_fma(t2, sj, jk[j][2]);
            }
        }
    }
}

// Acceleration only.  Follows GravitationAcc (swarm/gpu/gravitation_acc.hpp:57-74,96-139):
// `planets_only` skips every pair that touches body 0 (sum_acc_planets, for MVS).
// r4 (optional) receives |dx * r^-3|^2 per pair = what Verlet's one_over_r() returns (L78-93).
template <int N, bool PLANETS_ONLY>
__device__ __forceinline__ void acc_only(const double (&p)[N][3], const double (&m)[N],
                                         double (&a)[N][3]) {
#pragma unroll
    for (int i = 0; i < N; i++) a[i][0] = a[i][1] = a[i][2] = 0.0;
#pragma unroll
    for (int i = PLANETS_ONLY ? 1 : 0; i < N - 1; i++) {
#pragma unroll
        for (int j = i + 1; j < N; j++) {
            const double dx0 = p[j][0] - p[i][0], dx1 = p[j][1] - p[i][1], dx2 = p[j][2] - p[i][2];
            const double r2 = __builtin_fma(dx2, dx2, __builtin_fma(dx1, dx1, dx0 * dx0));
            const double y = rsqrt_full(r2);
            const double rinv = y * (y * y);
            const double si = rinv * m[j];
            const double sj = -(rinv * m[i]);
            a[i][0] = __builtin_fma(dx0, si, a[i][0]);
            a[i][1] = __builtin_fma(dx1, si, a[i][1]);
            a[i][2] = __builtin_fma(dx2, si, a[i][2]);
            a[j][0] = __builtin_fma(dx0, sj, a[j][0]);
            a[j][1] = __builtin_fma(dx1, sj, a[j][1]);
            a[j][2] = __builtin_fma(dx2, sj, a[j][2]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ghost lanes.  A fixed-step kernel with nothing to monitor lets the lanes whose systems are not ACTIVE sit the launch
// out.  Where that leaves 8 or fewer lanes of a wavefront enabled, FP64 instructions issue at a third of their rate
// (DESIGN 3.11): enough of the idle lanes then ride along as ghosts -- copies of the wave's active system with the
// earliest time (the one that takes the most steps), so that they do exactly what that lane does and bring no
// divergence of their own; they store nothing at the end.  Called by every lane of the wavefront that holds a system;
// returns whether this lane is a ghost.  A full wave of active systems pays one ballot.
template <int N>
__device__ __forceinline__ bool take_ghosts(const bool mine, State<N> &s) {
    const unsigned long long active = __ballot(mine);
    const int nlive = __popcll(active);
    if (nlive == 0 || nlive > 8) return false; // wave-uniform
    const unsigned long long idle = __ballot(!mine);
    const int rank = __popcll(idle & ((1ull << (threadIdx.x % W)) - 1)); // my place among the idle lanes
    int src = __ffsll((long long)active) - 1;
    double t0 = __shfl(s.t, src);
    for (unsigned long long m = active & (active - 1); m; m &= m - 1) { // at most 7 more
        const int l = __ffsll((long long)m) - 1;
        const double tl = __shfl(s.t, l);
        if (tl < t0) t0 = tl, src = l;
    }
    const bool ghost = !mine && rank < 9 - nlive;
#pragma unroll
    for (int b = 0; b < N; b++) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double x = __shfl(s.p[b][c], src), y = __shfl(s.v[b][c], src);
            if (ghost) s.p[b][c] = x, s.v[b][c] = y;
        }
        const double m = __shfl(s.m[b], src);
        if (ghost) s.m[b] = m;
    }
    if (ghost) s.t = t0, s.state = SWARM_HIP_ACTIVE;
    return ghost;
}

// ---------------------------------------------------------------------------------------------
// Lane-per-body kernels (one wavefront per workgroup): K doubles per lane handed to the other lanes of the wavefront
// through LDS -- one or two wide stores per lane and K/2 16-byte loads per partner instead of 2 K ds_bpermute_b32.
// The exchange is what bounds those kernels at saturation (irk2, 65536 x 6: 8.29 -> 7.75 ms; 2048 x 6: 1.135 -> 1.085).
// `slot` is a __shared__ array of W rows declared by the kernel; put() ... get() ... done() bracket one exchange.
template <int K>
struct LaneExchange {
    static constexpr int KP = (K + 1) / 2 * 2; // rows padded to 16 bytes
    typedef double Row[KP];
    static __device__ __forceinline__ void put(Row *slot, const int lane, const double (&x)[K]) {
#pragma unroll
        for (int k = 0; k < K; k++) slot[lane][k] = x[k];
        __syncthreads(); // a single wavefront: orders the stores before the loads, no waiting
    }
    static __device__ __forceinline__ void get(const Row *slot, const int src, double (&x)[K]) {
#pragma unroll
        for (int k = 0; k < K; k++) x[k] = slot[src][k];
    }
    static __device__ __forceinline__ void done() { __syncthreads(); } // loads before the next put()
};

// Lane-per-body kernels, the two forms of the loops over a body's partners.  UNROLLED in full, the partners' masses in
// nbod - 1 registers: every LDS load of an exchange is in flight at once, which is what a lone wavefront on its SIMD
// needs, at the price of registers (nbod 13: 256 VGPRs + 29 ... 98 AGPRs, one wavefront per SIMD whatever the ensemble
// size; nbod 32: 256 + 256 + 1.0 ... 1.4 KB of scratch per lane, and sixteen such instantiations of every kernel would
// take the library from 35 to 150 MB).  ROLLED, eight or sixteen partners per trip, the masses read from a row in LDS:
// half the registers, so several wavefronts share a SIMD and hide one another's LDS latency.  Up to BODY_UNROLL_MAX the kernels
// are unrolled (instruction for instruction what they were before the rolled form existed), beyond it (17 .. 32) rolled;
// the Hermite kernels are built in BOTH forms for nbod 12 .. 16, MVS for 14 .. 16, euler, midpoint and hermite_prop for
// 13 .. 16, where the rolled form is 10 - 30 % ahead once there is more than about a wavefront and a half per SIMD
// (profiles/r3/body_rolled_sweep_nbod_11_16.txt, prop_rolled_sweep.txt) and the launcher picks by the number of
// ACTIVE systems; rkck, irk2 and verlet lose with it at every size and are not.
// Both forms perform the same operations in the same order: bit-identical results (tests/test_gpu_nbod_11_32.py).
// SWH_BODY_LOOP goes before such a loop; it expects `BODY_TRIP` (= BodyLoops<N>::trip(ROLLED)) in scope.
constexpr int BODY_UNROLL_MAX = 16;
template <int N>
struct BodyLoops {
    static constexpr bool ROLLED = N > BODY_UNROLL_MAX; // the form of a kernel that is built in one form only
    // partners per trip of a rolled loop: 8, or 16 where a kernel is ahead with it (rkck, irk2, verlet: 2 - 10 %,
    // at nbod 17 -- one trip -- 25 %; the others lose up to 30 % there; profiles/r3/body_rolled_trip_8_vs_16.txt)
    static constexpr int trip(const bool rolled, const int partners = 8) { return rolled ? partners : N; }
    static constexpr int outer_trip(const bool rolled) { return rolled ? 1 : N; } // around a SWH_BODY_LOOP
};
#define SWH_BODY_LOOP _Pragma("unroll BODY_TRIP")
#define SWH_BODY_LOOP_OUTER _Pragma("unroll BODY_TRIP_OUTER")

// ---------------------------------------------------------------------------------------------
// Per-step constants that depend only on h.  For the fixed-step integrator h changes at most once
// per launch (the clipped last step), so they are recomputed only when h changes.
struct StepConsts {
    double h, hh, h3;        // h, h*0.5, h/3.0             (predictor, hermite_integrator.hpp:131-132)
    double kp_a, kp_j0, kp_j1; // position corrector weights  (L146)
    double kv_a, kv_j0, kv_j1; // velocity corrector weights  (L147)
    // FAST: h/3 through the 1-ulp reciprocal path (adaptive steps recompute the constants every step and
    // an IEEE divide costs ~60 cycles there); the fixed-step kernels keep the IEEE quotient
    template <bool FAST = false>
    __device__ __forceinline__ void set(double h_) {
        h = h_;
        hh = h_ * 0.5;
        h3 = FAST ? h_ * (1.0 / 3.0) : h_ / 3.0;
        const double h2 = h_ * h_;
        const double c60 = (1.0 / 60.0) * h_, c12 = (1.0 / 12.0) * h_;
        // pos = pre + ((0.1-0.25)*(a0-a1) - 1/60*(7 j0 + 2 j1) h) h^2, expanded per operand
        kp_a = (0.1 - 0.25) * h2;
        kp_j0 = -(7.0 * c60) * h2;
        kp_j1 = -(2.0 * c60) * h2;
        // vel = pre + ((-0.5)*(a0-a1) - 1/12*(5 j0 + j1) h) h
        kv_a = (-0.5) * h_;
        kv_j0 = -(5.0 * c12) * h_;
        kv_j1 = -c12 * h_;
    }
};

// One PEC2 Hermite step on register-resident state: predictor, 2 x (force + corrector).
// (a0,j0) = forces at the start of the step, (a1,j1) receive the forces at the end; the caller
// alternates the two buffer pairs so no copy `acc0 = acc1` (hermite_integrator.hpp:163) is needed.
// The corrector is expanded per operand, with the a0,j0 terms formed once per step.
// POSITION follows hermite_cpu's un-factored form `pre_pos + c1*(a0-a1)*h*h - c2*(7 j0+2 j1)*h*h*h`
// (integrators/hermite_cpu.hpp:152-158): the acceleration part and the jerk part are added to
// pre_pos SEPARATELY.  They cancel to O(h^4), i.e. far below one ulp of pos for small h, so the
// CUDA kernel's factored form `pre_pos + (A - B)` (hermite_integrator.hpp:146) always rounds the
// net correction away (stagnation): measured with the oracle at dt=1e-4, r=1, 1e5 steps it
// drifts 6.6e-12 in position and conserves energy 10x worse (5.8e-13 vs 4.1e-14) than
// hermite_cpu, the pinned reference.  Adding the parts one after the other keeps the roundings unbiased.
// Round 3: EVERY one of the four terms (c1 h^2 a0, -c1 h^2 a1, and the two jerk terms) is added to the position by
// an fma of its own -- `fma(kj1, j1, fma(-ka, a1, pp))` with pp = fma(kj0, j0, fma(ka, a0, pre_pos)) -- instead of
// `fma(-ka, a1, pp) + fma(kj1, j1, kj0*j0)`: one operation per coordinate and corrector pass fewer and nine doubles
// fewer live across the force evaluations (headline 16.83 -> 15.81 ms, 432 -> 414 instructions per step; nbod 6 0.68 ->
// 0.75 of peak).  Energy conservation is what it was (tools/exp_hermite_energy.py, profiles/r3/hermite_corrector_forms.txt:
// trivial.py's BasicIntegration case 4.5e-15 against 5.5e-15, the reference's bar is 1e-13; 4096 perturbed systems
// median 1.74e-14 against 1.68e-14), the distance from hermite_cpu after 1e5 steps 1.1e-12 against 9.0e-13.
// VELOCITY keeps its single rounding: the correction is summed first and added once.  The same treatment of the
// velocity (four fmas onto pre_vel) is another 4 % faster (15.14 ms) but puts four roundings of ulp(v) into every
// step, and BasicIntegration then conserves energy to 2.1e-13 -- beyond the reference's own bar; not taken.
// `vp` = the velocity the POSITION predictor uses: the reference kernel predicts from the global
// pos/vel but with its register copy of vel (hermite_integrator.hpp:124-132); the two differ only
// after a monitor rewrote the system (mce_stat merges).
constexpr int HERMITE_FOLD_FROM = 9;
template <int N>
__device__ __forceinline__ void hermite_step(State<N> &s, const StepConsts &k, const double (&a0)[N][3],
                                             const double (&j0)[N][3], double (&a1)[N][3],
                                             double (&j1)[N][3], const double (&vp)[N][3]) {
    // FOLD (nbod >= 9): the start-of-step part of the velocity correction is added to the predicted velocity once
    // instead of being kept beside it: 3N doubles fewer live across the two force evaluations (250 -> 220 at nbod 10),
    // at the price of one more rounding per step in v.  Fixed-step kernel, 65536 systems: nbod 9 132 B of scratch -> 0
    // (0.56 -> 0.66 of FP64 peak), nbod 10 384 -> 220 B (0.38 -> 0.48).  (Keeping pp and c0 in LDS instead was tried:
    // the scheduler hoists the reads above the force evaluation and spills more than before, volatile or not.)
    constexpr bool FOLD = N >= HERMITE_FOLD_FROM;
    double pp[N][3], pv[N][3], d0[FOLD ? 1 : N][3];
#pragma unroll
    for (int b = 0; b < N; b++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            pp[b][c] = __builtin_fma(k.h, __builtin_fma(k.hh, __builtin_fma(k.h3, j0[b][c], a0[b][c]), vp[b][c]),
                                     s.p[b][c]);
            pv[b][c] = __builtin_fma(k.h, __builtin_fma(k.hh, j0[b][c], a0[b][c]), s.v[b][c]);
            s.p[b][c] = pp[b][c];
            s.v[b][c] = pv[b][c];
            // pre_pos + c1*h^2*a0, then the start-of-step jerk term: every term of the position corrector is added to
            // the position by an fma of its own
            pp[b][c] = __builtin_fma(k.kp_j0, j0[b][c], __builtin_fma(k.kp_a, a0[b][c], pp[b][c]));
            const double dd = __builtin_fma(k.kv_a, a0[b][c], k.kv_j0 * j0[b][c]);
            if (FOLD)
                pv[b][c] = pv[b][c] + dd;
            else
                d0[b][c] = dd;
        }
#pragma unroll
    for (int round = 0; round < 2; round++) {
        accjerk<N>(s.p, s.v, s.m, a1, j1);
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                s.p[b][c] = __builtin_fma(k.kp_j1, j1[b][c], __builtin_fma(-k.kp_a, a1[b][c], pp[b][c]));
                if (FOLD)
                    s.v[b][c] = pv[b][c] + __builtin_fma(-k.kv_a, a1[b][c], k.kv_j1 * j1[b][c]);
                else
                    s.v[b][c] = pv[b][c] + __builtin_fma(-k.kv_a, a1[b][c], __builtin_fma(k.kv_j1, j1[b][c], d0[b][c]));
            }
    }
}

template <int N>
__device__ __forceinline__ void hermite_step(State<N> &s, const StepConsts &k, const double (&a0)[N][3],
                                             const double (&j0)[N][3], double (&a1)[N][3],
                                             double (&j1)[N][3]) {
    double vp[N][3];
#pragma unroll
    for (int b = 0; b < N; b++)
#pragma unroll
        for (int c = 0; c < 3; c++) vp[b][c] = s.v[b][c];
    hermite_step<N>(s, k, a0, j0, a1, j1, vp);
}

// ---------------------------------------------------------------------------------------------
// Device event buffer.  Record bytes = reference EVT_SNAPSHOT (swarm/log/log.hpp:313-326,
// log/types.hpp:37-56, gpulog/bits/gpulog_msg_layout.h:112-204).  The reference reserves with
// one atomicAdd per record on a single int (bits/gpulog_write.h:14-22); here the lanes of a
// wavefront that log in the same step reserve together: one atomic per wavefront.
__device__ __forceinline__ int snapshot_len(int nbod) { return (32 + 64 * nbod + 15) / 16 * 16; }

template <int N>
__device__ __forceinline__ void snapshot_fill(uint8_t *r, double t, int id, int state, const double (&p)[N][3],
                                              const double (&v)[N][3], const double (&m)[N]) {
    const int len = snapshot_len(N);
    int32_t *ri = reinterpret_cast<int32_t *>(r);
    double *rd = reinterpret_cast<double *>(r);
    ri[0] = 1; // EVT_SNAPSHOT
    ri[1] = len;
    rd[1] = t;
    ri[4] = id;
    ri[5] = state;
    ri[6] = N;
    ri[7] = 0;
#pragma unroll
    for (int b = 0; b < N; b++) {
        double *q = rd + 4 + 8 * b;
        q[0] = p[b][0], q[1] = p[b][1], q[2] = p[b][2];
        q[3] = v[b][0], q[4] = v[b][1], q[5] = v[b][2];
        q[6] = m[b];
        reinterpret_cast<int32_t *>(q + 7)[0] = b;
        reinterpret_cast<int32_t *>(q + 7)[1] = 0;
    }
    for (int k = (32 + 64 * N) / 8; k < len / 8; k++) rd[k] = 0.0; // pad to 16
}

template <int N>
__device__ __forceinline__ void log_system(const LogBuf &lg, const State<N> &s, bool want) {
    const int len = snapshot_len(N);
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    const int cnt = __popcll(mask);
    int base = 0;
    if (lane == leader && lg.buf) base = atomicAdd(lg.cursor, cnt * len);
    base = __shfl(base, leader);
    if (!want) return;
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    const long long off = (long long)base + (long long)rank * len;
    if (!lg.buf || off + len > (long long)lg.cap) {
        // overflow: give the bytes back and count the drop (the reference drops silently)
        if (lg.buf) atomicSub(lg.cursor, len);
        if (lg.cursor) atomicAdd(lg.cursor + 1, 1);
        return;
    }
    snapshot_fill<N>(lg.buf + off, s.t, s.id, s.state, s.p, s.v, s.m);
}

// The same record written by the N lanes of a system in the lane-per-body kernels: the star's lanes of the systems
// that log in this step reserve together (one atomic per wavefront), every lane writes its own body's 64 bytes and the
// star's lane the 32-byte header.  Called by the whole wavefront; `want` is the same in all lanes of a system,
// `star` = the lane of the system's body 0, `b` = my body.
template <int N>
__device__ __forceinline__ void log_system_body(const LogBuf &lg, const bool want, const int b, const int star,
                                                const double t, const int id, const int state, const double (&p)[3],
                                                const double (&v)[3], const double m) {
    const int len = snapshot_len(N);
    const unsigned long long mask = __ballot(want && b == 0);
    if (mask == 0) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    const int cnt = __popcll(mask);
    int base = 0;
    if (lane == leader && lg.buf) base = atomicAdd(lg.cursor, cnt * len);
    base = __shfl(base, leader);
    if (!want) return;
    const int rank = __popcll(mask & ((1ull << star) - 1ull));
    const long long off = (long long)base + (long long)rank * len;
    if (!lg.buf || off + len > (long long)lg.cap) {
        // overflow: give the bytes back and count the drop (the reference drops silently)
        if (b == 0) {
            if (lg.buf) atomicSub(lg.cursor, len);
            if (lg.cursor) atomicAdd(lg.cursor + 1, 1);
        }
        return;
    }
    uint8_t *r = lg.buf + off;
    double *rd = reinterpret_cast<double *>(r);
    if (b == 0) {
        int32_t *ri = reinterpret_cast<int32_t *>(r);
        ri[0] = 1; // EVT_SNAPSHOT
        ri[1] = len;
        rd[1] = t;
        ri[4] = id;
        ri[5] = state;
        ri[6] = N;
        ri[7] = 0;
        for (int k = (32 + 64 * N) / 8; k < len / 8; k++) rd[k] = 0.0; // pad to 16
    }
    double *q = rd + 4 + 8 * b;
    q[0] = p[0], q[1] = p[1], q[2] = p[2];
    q[3] = v[0], q[4] = v[1], q[5] = v[2];
    q[6] = m;
    reinterpret_cast<int32_t *>(q + 7)[0] = b;
    reinterpret_cast<int32_t *>(q + 7)[1] = 0;
}

// Per-lane reservation (bits/gpulog_write.h:14-22) for the rare records written from divergent
// code: lprintf lines and the snapshots mce_stat writes in the middle of its pass.
__device__ __forceinline__ uint8_t *log_reserve_lane(const LogBuf &lg, int len) {
    if (!lg.buf) {
        if (lg.cursor) atomicAdd(lg.cursor + 1, 1);
        return nullptr;
    }
    const int off = atomicAdd(lg.cursor, len);
    if ((long long)off + len > (long long)lg.cap) {
        atomicSub(lg.cursor, len);
        atomicAdd(lg.cursor + 1, 1);
        return nullptr;
    }
    return lg.buf + off;
}

// lprintf = log.write(MSG_PRINTF, fmt, args...) (gpulog/bits/gpulog_printf.h, lprintf.h:47-58).
// Record: header{msgid=-2 (gpulog_constants.h:35), len} | char fmt[n] at 8, NUL included | each
// argument at its natural alignment (int32: 4, double: 8) | padded to 16
// (bits/gpulog_msg_layout.h:112-204).  The host replays them with run_printf (lprintf.h:74-171).
struct PArg {
    int is_double;
    int i;
    double d;
};
__device__ __forceinline__ PArg parg(int v) { return PArg{0, v, 0.0}; }
__device__ __forceinline__ PArg parg(double v) { return PArg{1, 0, v}; }

static __device__ __noinline__ void log_printf_lane(const LogBuf &lg, const char *fmt, int n, const PArg *a, int nargs) {
    int at = 8 + n;
    for (int k = 0; k < nargs; k++) {
        const int al = a[k].is_double ? 8 : 4;
        at = (at + al - 1) / al * al + al;
    }
    const int len = (at + 15) / 16 * 16;
    uint8_t *r = log_reserve_lane(lg, len);
    if (!r) return;
    atomicAdd(lg.cursor + 2, 1);
    for (int k = 8; k < len; k += 4) *reinterpret_cast<int32_t *>(r + k) = 0;
    reinterpret_cast<int32_t *>(r)[0] = -2; // MSG_PRINTF
    reinterpret_cast<int32_t *>(r)[1] = len;
    for (int k = 0; k < n; k++) r[8 + k] = (uint8_t)fmt[k];
    at = 8 + n;
    for (int k = 0; k < nargs; k++) {
        const int al = a[k].is_double ? 8 : 4;
        at = (at + al - 1) / al * al;
        if (a[k].is_double)
            *reinterpret_cast<double *>(r + at) = a[k].d;
        else
            *reinterpret_cast<int32_t *>(r + at) = a[k].i;
        at += al;
    }
}
// The same record written in line: no call, no argument array on the stack.  For the rare lines of the body monitors
// (body_monitor.hpp): a call in a lane-per-body kernel costs its hot loop 4 - 9 % even if it is never taken
// (profiles/r3/body_monitor_print_path_ab.txt).  The loops over the text stay rolled.
template <int NFMT, int NARGS>
__device__ __forceinline__ void log_printf_inline(const LogBuf &lg, const char (&fmt)[NFMT], const PArg (&a)[NARGS]) {
    int at = 8 + NFMT;
#pragma unroll
    for (int k = 0; k < NARGS; k++) {
        const int al = a[k].is_double ? 8 : 4;
        at = (at + al - 1) / al * al + al;
    }
    const int len = (at + 15) / 16 * 16;
    uint8_t *r = log_reserve_lane(lg, len);
    if (!r) return;
    atomicAdd(lg.cursor + 2, 1);
#pragma unroll 1
    for (int k = 8; k < len; k += 4) *reinterpret_cast<int32_t *>(r + k) = 0;
    reinterpret_cast<int32_t *>(r)[0] = -2; // MSG_PRINTF
    reinterpret_cast<int32_t *>(r)[1] = len;
#pragma unroll 1
    for (int k = 0; k < NFMT; k++) r[8 + k] = (uint8_t)fmt[k];
    at = 8 + NFMT;
#pragma unroll
    for (int k = 0; k < NARGS; k++) {
        const int al = a[k].is_double ? 8 : 4;
        at = (at + al - 1) / al * al;
        if (a[k].is_double)
            *reinterpret_cast<double *>(r + at) = a[k].d;
        else
            *reinterpret_cast<int32_t *>(r + at) = a[k].i;
        at += al;
    }
}
#define SWH_LPRINTF_INLINE(lg, fmt, ...)           \
    do {                                           \
        const PArg _a[] = {__VA_ARGS__};           \
        log_printf_inline(lg, fmt, _a);            \
    } while (0)

#define SWH_LPRINTF(lg, fmt, ...)                                  \
    do {                                                           \
        const PArg _a[] = {__VA_ARGS__};                           \
        log_printf_lane(lg, fmt, (int)sizeof(fmt), _a, (int)(sizeof(_a) / sizeof(_a[0]))); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Monitors.  Every comparison that decides an integer outcome is evaluated without FMA
// contraction, in the reference's operation order.
struct MonState {
    bool ej_met, ce_met;
    double next_log_time;
};

template <int N>
__device__ __forceinline__ void mon_init(MonState &ms, const State<N> &s) {
    ms.ej_met = false;
    ms.ce_met = false;
    ms.next_log_time = s.t; // monitors/log_time_interval.hpp:118-119
}

// The first two early-outs of test_body (stop_on_ejection.hpp:112-117) without the square root: false means that the
// body cannot meet the condition -- it is inside rmax for certain (r2 below a host-computed strict lower bound of
// rmax^2) or not moving outwards (the reference's own `rdotv <= 0` test, same operations).  The kernels evaluate it for
// every planet on every step and reach the monitor proper only when some lane of the wavefront answers true: with the
// second test a planet that sits beyond rmax on a bound orbit (rmax inside the outermost orbit) stops costing a
// monitor call per step.
__device__ __forceinline__ bool ej_maybe_body(const double (&p)[3], const double (&v)[3], const MonParams &mp) {
#pragma clang fp contract(off)
    const double r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    const double rdotv = p[0] * v[0] + p[1] * v[1] + p[2] * v[2];
    return !(r2 < mp.r2_below) && !(rdotv <= 0.);
}

// stop_on_ejection::test_body, monitors/stop_on_ejection.hpp:108-138
// VERBOSE (compile time) = the kernel variant that can emit lprintf records: the out-of-line
// formatter call costs registers in the step loop, so launches without verbose_on_* flags use
// the variant without it.
template <int N, bool VERBOSE>
__device__ __forceinline__ bool ej_test_body(const State<N> &s, int b, const MonParams &mp, const LogBuf &lg) {
#pragma clang fp contract(off)
    const double x = s.p[b][0], y = s.p[b][1], z = s.p[b][2];
    const double r2 = x * x + y * y + z * z;
    if (r2 < mp.r2_below) return false; // certainly r < rmax; skips the sqrt on the common path
    const double vx = s.v[b][0], vy = s.v[b][1], vz = s.v[b][2];
    const double r = sqrt(r2);
    if (r < mp.rmax) return false;
    const double rdotv = x * vx + y * vy + z * vz;
    if (rdotv <= 0.) return false;
    const double speed_sq = vx * vx + vy * vy + vz * vz;
    const double epp = 0.5 * speed_sq * r / s.m[b] - 1.;
    const bool parabolic = fabs(epp) < 1e-4;
    if (parabolic || epp > 0) {
        if (VERBOSE && mp.ej_verbose) { // L122-130
            const double energy = 0.5 * speed_sq - s.m[b] / r;
            if (parabolic)
                SWH_LPRINTF(lg, "Orbit is nearly parabolic: _sys=%d, bod=%d, T=%lg r=%lg energy=%lg energy*r/GM=%lg.\n",
                            parg(s.number), parg(b), parg(s.t), parg(r), parg(energy), parg(epp));
            else
                SWH_LPRINTF(lg, "Orbit is hyperbolic: _sys=%d, bod=%d, T=%lg r=%lg energy=%lg energy*r/GM=%lg.\n",
                            parg(s.number), parg(b), parg(s.t), parg(r), parg(energy), parg(epp));
        }
        return true;
    }
    return false;
}

// test_body without the verbose branch, on one body's scalars; same operations as ej_test_body
__device__ __forceinline__ bool ej_test_quiet(const double (&p)[3], const double (&v)[3], const double m,
                                              const MonParams &mp) {
#pragma clang fp contract(off)
    const double x = p[0], y = p[1], z = p[2];
    const double r2 = x * x + y * y + z * z;
    if (r2 < mp.r2_below) return false;
    const double vx = v[0], vy = v[1], vz = v[2];
    const double r = sqrt(r2);
    if (r < mp.rmax) return false;
    const double rdotv = x * vx + y * vy + z * vz;
    if (rdotv <= 0.) return false;
    const double speed_sq = vx * vx + vy * vy + vz * vz;
    const double epp = 0.5 * speed_sq * r / m - 1.;
    return fabs(epp) < 1e-4 || epp > 0;
}

// stop_on_close_encounter::check_close_encounters, monitors/stop_on_close_encounter.hpp:124-142, for one pair on
// scalars: bodies i and j, the star's mass GM; `d` receives the separation
__device__ __forceinline__ bool ce_test_quiet(const double (&pi)[3], const double (&pj)[3], const double mi,
                                              const double mj, const double GM, const MonParams &mp, double &d) {
#pragma clang fp contract(off)
    const double dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
    d = sqrt(dx * dx + dy * dy + dz * dz);
    const double ri = sqrt(pi[0] * pi[0] + pi[1] * pi[1] + pi[2] * pi[2]);
    const double a = 0.5 * (ri + ri); // the reference uses body i twice (L130)
    const double rH3 = (mi + mj) / (3. * GM) * a * a * a;
    return d * d * d < mp.dmin3 * rH3;
}
template <int N, bool VERBOSE>
__device__ __forceinline__ bool ce_check(const State<N> &s, int i, int j, const MonParams &mp, const LogBuf &lg) {
    double d;
    const bool close_encounter = ce_test_quiet(s.p[i], s.p[j], s.m[i], s.m[j], s.m[0], mp, d);
    if (VERBOSE && close_encounter && mp.ce_verbose) // L137-141
        SWH_LPRINTF(lg, "Close apporach detected: sys=%d, T=%f j=%d i=%d  d=%lg.\n", parg(s.number), parg(s.t),
                    parg(j), parg(i), parg(d));
    return close_encounter;
}

// pass_one of the bound monitor (stop_on_ejection.hpp:177-193, log_time_interval.hpp:98-111,
// stop_on_close_encounter.hpp:101-115, composites.hpp:199-200).  Returns need_full_test
// ("this system needs standard coordinates").
template <int N, bool VERBOSE>
__device__ __forceinline__ bool mon_pass_one(MonState &ms, const State<N> &s, const MonParams &mp,
                                             const LogBuf &lg) {
    bool need_full = false;
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) {
        ms.ej_met = false;
        if (mp.log_interval != 0.) {
            if (s.t >= ms.next_log_time) {
                ms.ej_met = true;
                ms.next_log_time += mp.log_interval;
            }
        }
        return ms.ej_met;
    }
    ms.ej_met = false;
    if (mp.ej_any) {
        // One wave-uniform branch per call instead of one divergent branch per planet: at one wave per SIMD a
        // taken branch stalls the wave while its instruction buffer refills.  `maybe` is the negation of the
        // first early-out of test_body for any planet; the exact tests run only when some lane needs them.
        bool maybe = false;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int b = 1; b < N; b++) {
                maybe = maybe || ej_maybe_body(s.p[b], s.v[b], mp);
            }
        }
        if (__any(maybe)) {
#pragma unroll
            for (int b = 1; b < N; b++) ms.ej_met = ms.ej_met || ej_test_body<N, VERBOSE>(s, b, mp, lg);
        }
        if (ms.ej_met && mp.ej_log) need_full = true;
    }
    if (mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE && !need_full) { // short-circuit ||
        ms.ce_met = false;
        if (mp.ce_any) {
#pragma unroll
            for (int b = 2; b < N; b++)
#pragma unroll
                for (int d = 1; d < b; d++) ms.ce_met = ms.ce_met || ce_check<N, VERBOSE>(s, b, d, mp, lg);
            if (ms.ce_met && mp.ce_log) need_full = true;
        }
    }
    return need_full;
}

// pass_two: stop_on_ejection.hpp:196-201, stop_on_close_encounter.hpp:118-123
template <int N>
__device__ __forceinline__ void mon_pass_two(const MonState &ms, State<N> &s, const MonParams &mp) {
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) return;
    if (ms.ej_met && mp.ej_deactivate) s.state = SWARM_HIP_DISABLED;
    if (mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE && ms.ce_met && mp.ce_deactivate)
        s.state = SWARM_HIP_DISABLED;
}

__device__ __forceinline__ bool mon_need_log(const MonState &ms, const MonParams &mp) {
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) return mp.log_interval != 0. && ms.ej_met;
    bool r = mp.ej_log && ms.ej_met;
    if (mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE) r = r || (mp.ce_log && ms.ce_met);
    return r;
}

// Monitor::operator()(thread_in_system): pass_one, pass_two, log (stop_on_ejection.hpp:140-146).
// `participate` = this lane really calls the monitor.  The lanes that reach this function together
// (the whole wave in rkck, the lanes still integrating elsewhere) reserve log space together: the
// ballots inside see exactly the lanes that are here.
template <int N, bool VERBOSE>
__device__ __forceinline__ void mon_call(MonState &ms, State<N> &s, const MonParams &mp,
                                         const LogBuf &lg, bool participate) {
    bool want = false;
    if (mp.kind == SWARM_HIP_MON_EJECTION) {
        // Common case of the common monitor, decided with two scalar branches: no planet of any lane of the
        // wave is beyond rmax (r2 < r2_below is the first early-out of test_body), so nothing is met,
        // nothing is logged, the state is unchanged.  Otherwise fall through to the general path.
        bool maybe = false;
        if (participate && mp.ej_any) {
#pragma clang fp contract(off)
#pragma unroll
            for (int b = 1; b < N; b++) {
                maybe = maybe || ej_maybe_body(s.p[b], s.v[b], mp);
            }
        }
        if (!__any(maybe)) {
            ms.ej_met = false;
            return;
        }
    }
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) { // same early-out: no lane of the wave is due for a snapshot
        const bool due = participate && mp.log_interval != 0. && s.t >= ms.next_log_time;
        if (!__any(due)) {
            ms.ej_met = false;
            return;
        }
    }
    if (participate) {
        mon_pass_one<N, VERBOSE>(ms, s, mp, lg);
        mon_pass_two(ms, s, mp);
        want = mon_need_log(ms, mp);
    }
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL || mp.ej_log || mp.ce_log)
        if (__any(want)) log_system(lg, s, want);
}

// ---------------------------------------------------------------------------------------------
// log_transit (monitors/log_transit.hpp, plugin hermite_adap_transit, plugins/hermite_adap.cu:61):
// logs EVT_TRANSIT / EVT_OCCULTATION events (estimated mid-time, impact parameter and sky-projected
// speed in stellar radii) when a planet passes in front of / behind body 0 as seen along +z, plus a
// snapshot of the system.  The reference is compiled with USE_WORKS 1 (L32): the estimate is the
// Taylor extrapolation of calc_transit_time_works (L213-253) from the accelerations and jerks of the
// integrator's LAST force evaluation, which it reads back from the gravitation object's shared
// memory -- here the (a, jk) arrays the Hermite step has just produced.  On its first call, before
// any force evaluation (hermite_adap.hpp:130), the reference reads that shared memory
// uninitialised; here the first call sees the forces of the initial state.  Marked EXPERIMENTAL in
// the reference; parity unpinned.
//
// log::event(log, evt, T, sys, j, b, vproj) (log/log.hpp:261-265): header | T @8 | sys @16 | j @20 |
// b @24 | vproj @32, 40 bytes padded to 48.
__device__ __forceinline__ void log_transit_event(const LogBuf &lg, int evt, double T, int sys, int j, double b,
                                                  double vproj) {
    uint8_t *r = log_reserve_lane(lg, 48);
    if (!r) return;
    int32_t *ri = reinterpret_cast<int32_t *>(r);
    double *rd = reinterpret_cast<double *>(r);
    ri[0] = evt, ri[1] = 48;
    rd[1] = T;
    ri[4] = sys, ri[5] = j;
    rd[3] = b, rd[4] = vproj, rd[5] = 0.0;
}

// check_in_transit(thread, i = 0, j, dt), L256-386
template <int N>
__device__ __forceinline__ bool transit_check(const State<N> &s, const int j, const double (&a)[N][3],
                                              const double (&jk)[N][3], const double (&radius)[N],
                                              const bool have_radius, const MonParams &mp, const LogBuf &lg) {
#pragma clang fp contract(off)
    const int i = 0;
    const double dt = mp.tr_dt;
    const double depth = s.p[j][2] - s.p[i][2];
    if (!mp.tr_log_occultation && (depth < 0.)) return false;
    if (!mp.tr_log_transit && (depth > 0.)) return false;
    double dx[2] = {s.p[j][0] - s.p[i][0], s.p[j][1] - s.p[i][1]};
    double dv[2] = {s.v[j][0] - s.v[i][0], s.v[j][1] - s.v[i][1]};
    const double b2begin = dx[0] * dx[0] + dx[1] * dx[1];
    const double db2dt = 2. * (dx[0] * dv[0] + dx[1] * dv[1]);
    const double radi = have_radius ? radius[i] : 0.;
    const double radj = have_radius ? radius[j] : 0.;
    const double lim = (radi + radj) * (1. + mp.tr_tol);
    if (fmin(b2begin - 0.5 * dt * db2dt, b2begin + 0.5 * dt * db2dt) > lim * lim) return false;
    // calc_transit_time_works, L213-253
    double dt_min_b2, b, vproj;
    {
        const double da[2] = {a[j][0] - a[i][0], a[j][1] - a[i][1]};
        const double dj[2] = {jk[j][0] - jk[i][0], jk[j][1] - jk[i][1]};
        const double d2b2dt2 = 2. * (dv[0] * dv[0] + dv[1] * dv[1] + dx[0] * da[0] + dx[1] * da[1]);
        const double d3b2dt3 = 6. * (dv[0] * da[0] + dv[1] * da[1]) + 2. * (dx[0] * dj[0] + dx[1] * dj[1]);
        double dt_try = -db2dt / d2b2dt2; // vertex of the parabola
        dt_try = -db2dt / (d2b2dt2 + 0.5 * dt_try * d3b2dt3);
        if ((dt_try < -0.5 * dt) || (dt_try > 0.5 * dt)) {
            dt_min_b2 = dt_try, b = -1., vproj = -1.;
        } else {
            dt_min_b2 = 0.;
            double min_b2 = b2begin;
            dx[0] += dt_try * (dv[0] + dt_try * 0.5 * (da[0] + dt_try * dj[0] / 3.));
            dx[1] += dt_try * (dv[1] + dt_try * 0.5 * (da[1] + dt_try * dj[1] / 3.));
            const double b2 = (dx[0] * dx[0] + dx[1] * dx[1]);
            if (b2 < min_b2) min_b2 = b2, dt_min_b2 = dt_try;
            b = (min_b2 > 0) ? sqrt(min_b2) : -sqrt(-min_b2);
            dv[0] += dt_min_b2 * (da[0] + 0.5 * dt_min_b2 * dj[0]);
            dv[1] += dt_min_b2 * (da[1] + 0.5 * dt_min_b2 * dj[1]);
            vproj = sqrt(dv[0] * dv[0] + dv[1] * dv[1]);
        }
    }
    if (vproj >= 0.) {
        if (have_radius) {
            if (depth > 0.)
                b /= radi, vproj /= radi;
            else
                b /= radj, vproj /= radj;
        }
        if ((dt_min_b2 >= -0.5 * dt) && (dt_min_b2 < 0.5 * dt))
            log_transit_event(lg, (depth > 0.) ? 15 /*EVT_TRANSIT*/ : 16 /*EVT_OCCULTATION*/, s.t + dt_min_b2, s.id, j, b,
                              vproj);
        return true;
    }
    return false;
}

// log_transit::operator(), L108-114, 389-402: pass_two ORs check_in_transit over the planets
// (short-circuit), then the system snapshot when the condition is met
template <int N>
__device__ __forceinline__ void transit_monitor(const State<N> &s, const double (&a)[N][3], const double (&jk)[N][3],
                                                const double (&radius)[N], const bool have_radius,
                                                const MonParams &mp, const LogBuf &lg) {
    bool met = false;
#pragma unroll
    for (int b = 1; b < N; b++) met = met || transit_check<N>(s, b, a, jk, radius, have_radius, mp, lg);
    if (met) {
        uint8_t *r = log_reserve_lane(lg, snapshot_len(N));
        if (r) snapshot_fill<N>(r, s.t, s.id, s.state, s.p, s.v, s.m);
    }
}

// kernel variant for a launch: 0 = monitor has nothing to do, 1 = monitor, 2 = monitor + lprintf
//                 3 = mce_stat (kernels_mce.hip, kernels_irk2.hip), 4 = log_transit (hermite_adap_transit)
// MON_ON: stop_on_ejection / log_time_interval, early-out inline + monitor out of line.  MON_VERBOSE: the monitor
// inline and lprintf-capable; also used for the close-encounter composite, which has no cheap early-out.
// MON_EJECT_QUIET (fixed-step hermite): stop_on_ejection with nothing to log or print -- the test alone, inline.
enum { MON_OFF = 0, MON_ON = 1, MON_VERBOSE = 2, MON_MCE = 3, MON_TRANSIT = 4, MON_EJECT_QUIET = 5 };
__host__ __device__ inline bool monitor_is_noop(const MonParams &mp);
inline int monitor_variant(const MonParams &mp) {
    if (mp.kind == SWARM_HIP_MON_LOG_TRANSIT) return mp.tr_tol != 0. ? MON_TRANSIT : MON_OFF; // is_log_on(), L97
    if (monitor_is_noop(mp)) return MON_OFF;
    const bool ej = mp.kind != SWARM_HIP_MON_LOG_INTERVAL && mp.ej_verbose;
    const bool ce = mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE && mp.ce_verbose;
    return (ej || ce) ? MON_VERBOSE : MON_ON;
}

__host__ __device__ inline bool monitor_is_noop(const MonParams &mp) {
    if (mp.kind == SWARM_HIP_MON_LOG_INTERVAL) return mp.log_interval == 0.;
    return !mp.ej_any && !(mp.kind == SWARM_HIP_MON_EJECTION_OR_CLOSE && mp.ce_any);
}

} // namespace swh
