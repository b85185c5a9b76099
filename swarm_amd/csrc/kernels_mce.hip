// kernels_mce.hip -- `hermite` with the mce_stat monitor for gfx950.
//
// Replaces hermite< mce_stat<L>, GravitationAccJerk > (plugins/hermite.cu:40,
// integrators/hermite_integrator.hpp:74-187, monitors/mercury_mce_stat.hpp).  The monitor needs
// body attributes 0 (radius), 1 (rce) and 2 (rphysics) and system attribute 0 (energy lost in a
// merge); with the reference's default NUM_PLANET_ATTRIBUTES=1 those accesses run past the Body
// struct, so the defined behaviour -- nbattr >= 3 -- is what is built, and the host refuses
// anything else.  PARITY UNPINNED: the reference ships no fixture or CPU twin of this monitor.
#include "mce_monitor.hpp"

namespace swh {

template <int N>
__global__ __launch_bounds__(256) void hermite_mce_kernel(EnsView e, IntParams ip, MonParams mp, LogBuf lg,
                                                          Counters cnt) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= e.nsys) return;
    State<N> s;
    load_state(e, gid, s);
    MceData<N> md;
    mce_init<N>(e, gid, s, mp, md);

    unsigned long long steps = 0;
    bool rewritten = false;
    // The wave goes through its steps whole: a lane whose system is done -- end of the integration, a
    // halt after a collision, the iteration budget -- or was not ACTIVE at launch hands its state back at once and
    // then only keeps the wave company with steps of h = 0, unmonitored, never stored: MI355X issues FP64 instructions
    // at a third of the rate once 8 or fewer lanes of a wavefront are enabled, and a run of mce_stat ends with a few
    // surviving systems per wave.
    bool stored = false;
    if (__any(s.state == SWARM_HIP_ACTIVE)) {
        double fa[2][N][3], fj[2][N][3], vreg[N][3], cp[N][3], cv[N][3];
        accjerk<N>(s.p, s.v, s.m, fa[0], fj[0]);
#pragma unroll
        for (int b = 0; b < N; b++)
#pragma unroll
            for (int c = 0; c < 3; c++) vreg[b][c] = s.v[b][c];
        const double dest = ip.destination_time;
        StepConsts k;
        k.set(ip.time_step);
        int iter = 0;
        auto live = [&]() { return iter < ip.max_iterations && s.state == SWARM_HIP_ACTIVE; };
        while (__any(live())) {
            const bool on = live();
            if (!on && !stored) {
                if (steps) store_state(e, gid, s);
                if (rewritten) mce_store_masses<N>(e, gid, s);
                stored = true;
            }
            double h = ip.time_step;
            {
#pragma clang fp contract(off)
                if (s.t + h > dest) h = dest - s.t;
            }
            if (!on) h = 0.0;
            if (h != k.h) k.set(h);
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) cp[b][c] = s.p[b][c], cv[b][c] = s.v[b][c];
            // the position predictor uses the kernel's register copy of vel, which a merge does not
            // touch, and acc0/jerk0 stay stale for one step (hermite_integrator.hpp:124-132)
            hermite_step<N>(s, k, fa[0], fj[0], fa[1], fj[1], vreg);
#pragma unroll
            for (int b = 0; b < N; b++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    vreg[b][c] = s.v[b][c];
                    fa[0][b][c] = fa[1][b][c], fj[0][b][c] = fj[1][b][c];
                }
            if (on) {
                s.t += h;
                steps++;
                iter++;
                if (mce_monitor<N>(e, gid, s, cp, cv, md, mp.mce_time_step, lg)) rewritten = true;
                if (s.state == SWARM_HIP_ACTIVE) {
#pragma clang fp contract(off)
                    if (s.t > dest - 1e-12) s.state = SWARM_HIP_INACTIVE; // hermite_integrator.hpp:177-179
                }
            }
        }
    }
    if (!stored) {
        if (steps) store_state(e, gid, s);
        if (rewritten) mce_store_masses<N>(e, gid, s);
    }
    if (cnt.accepted) cnt.accepted[gid] += steps;
    if (cnt.attempted) cnt.attempted[gid] += steps;
}

int launch_hermite_mce(const LaunchCtx &c) {
    if (int rc = mce_check_attributes(c)) return rc;
    SWH_DISPATCH_NBOD(c.ens.nbod, hipLaunchKernelGGL((hermite_mce_kernel<N>), dim3(c.grid), dim3(c.block), 0,
                                                     c.stream, c.ens, c.ip, c.mp, c.lg, c.cnt));
    return check_launch("hermite/mce_stat kernel launch");
}

} // namespace swh
