// tools/mvs_phase_budget.hip -- per-phase instruction budget of one MVS step (nbod 8, one lane per system).
// Not part of the library: every phase of MVSPropagator::advance as the library's own device functions perform it
// (kernels_generic.hip is included as it is), each in a kernel of its own that loads a state, runs the phase once and
// stores the result into a second ensemble (so that the load/store-only kernel keeps its loads and stores).  tools/mvs_phase_budget.py compiles this to gfx950 assembly and counts the VALU instructions of
// each kernel (less those of the load/store-only kernel): the static budget that profiles/r3/mvs_phase_budget.txt
// records beside the PMC total of the real kernel.
#include "../swarm_amd/csrc/kernels_generic.hip"

namespace swh {

constexpr int NB = 8;

#define PHASE_PROLOGUE                                            \
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;        \
    if (gid >= e.nsys) return;                                    \
    State<NB> s;                                                  \
    load_state(e, gid, s);                                        \
    double acc[NB][3];                                            \
    _Pragma("unroll") for (int b = 0; b < NB; b++)                \
    _Pragma("unroll") for (int c = 0; c < 3; c++) acc[b][c] = e.sysd[(b * 3 + c) * 64 + gid];
#define PHASE_EPILOGUE                                            \
    _Pragma("unroll") for (int b = 0; b < NB; b++)                \
    _Pragma("unroll") for (int c = 0; c < 3; c++) o_.sysd[(b * 3 + c) * 64 + gid] = acc[b][c]; \
    store_state(o_, gid, s);

__global__ __launch_bounds__(256, 2) void phase_none(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    PHASE_EPILOGUE
}
__global__ __launch_bounds__(256, 2) void phase_drift(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    double mv[3];
    mvs_momentum(s, mv);
    mvs_drift(s, mv, ip.time_step, ip.min_time_step);
    PHASE_EPILOGUE
}
__global__ __launch_bounds__(256, 2) void phase_kick(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
#pragma unroll
    for (int b = 1; b < NB; b++)
#pragma unroll
        for (int c = 0; c < 3; c++) s.v[b][c] = __builtin_fma(ip.time_step, acc[b][c], s.v[b][c]);
    PHASE_EPILOGUE
}
__global__ __launch_bounds__(256, 2) void phase_acc_planets(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    acc_only<NB, true>(s.p, s.m, acc);
    PHASE_EPILOGUE
}
// the seven planets' Kepler drifts as the kernel groups them (3, 2, 2), every pass of every group taken or left by
// the wave-uniform test as in the kernel
__global__ __launch_bounds__(256, 2) void phase_kepler_all(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    KeplerConsts kc;
    kc.set_star(ip.max_time_step);
    kc.set_dt(ip.time_step);
    drift_kepler_all<NB, false, true>(s, kc);
    PHASE_EPILOGUE
}
// one planet's pieces: begin, one unconditional pass (y, series, pass), one conditional pass, end
__global__ __launch_bounds__(256, 2) void phase_kepler_begin1(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    KeplerConsts kc;
    kc.set_star(ip.max_time_step);
    kc.set_dt(ip.time_step);
    KeplerOrbit o;
    kepler_begin(o, s.p[1], s.v[1], kc);
    acc[0][0] = o.r0, acc[0][1] = o.inv_r0, acc[0][2] = o.alpha, acc[1][0] = o.sig0, acc[1][1] = o.foo, acc[1][2] = o.x;
    PHASE_EPILOGUE
}
template <bool COND>
__device__ __forceinline__ void one_pass(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    KeplerConsts kc;
    kc.smu = ip.max_time_step, kc.smudt = ip.min_time_step;
    KeplerOrbit o;
    o.r0 = acc[0][0], o.inv_r0 = acc[0][1], o.alpha = acc[0][2], o.sig0 = acc[1][0], o.foo = acc[1][1], o.x = acc[1][2];
    o.u = acc[2][0];
    o.live = true;
    bool go = true;
    if (COND) go = __any(kepler_more(o, 3));
    if (go) {
        double y[1], Sp[1], Cp[1];
        y[0] = kepler_y(o);
        SC_group<1>(y, Sp, Cp);
        kepler_pass(o, kc, y[0], Sp[0], Cp[0]);
    }
    acc[1][2] = o.x, acc[2][0] = o.u;
    PHASE_EPILOGUE
}
__global__ __launch_bounds__(256, 2) void phase_kepler_pass_uncond1(EnsView e, EnsView o_, IntParams ip) { one_pass<false>(e, o_, ip); }
__global__ __launch_bounds__(256, 2) void phase_kepler_pass_cond1(EnsView e, EnsView o_, IntParams ip) { one_pass<true>(e, o_, ip); }
__global__ __launch_bounds__(256, 2) void phase_kepler_end1(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    KeplerConsts kc;
    kc.smu = ip.max_time_step, kc.inv_smu = ip.min_time_step, kc.dt = ip.time_step;
    KeplerOrbit o;
    o.r0 = acc[0][0], o.inv_r0 = acc[0][1], o.alpha = acc[0][2], o.sig0 = acc[1][0], o.foo = acc[1][1], o.x = acc[1][2];
    double y[1], Sp[1], Cp[1];
    y[0] = kepler_y(o);
    SC_group<1>(y, Sp, Cp);
    kepler_end(o, kc, y[0], Sp[0], Cp[0], false, true, s.p[1], s.v[1]);
    PHASE_EPILOGUE
}
__global__ __launch_bounds__(256, 2) void phase_coords(EnsView e, EnsView o_, IntParams ip) {
    PHASE_PROLOGUE
    mvs_to_internal(s);
    PHASE_EPILOGUE
}

} // namespace swh
