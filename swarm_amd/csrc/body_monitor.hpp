// body_monitor.hpp -- the monitors a lane-per-body kernel serves (kernels_*_body.hip), written once: stop_on_ejection
// (stop_on_ejection.hpp:108-201), log_time_interval (log_time_interval.hpp:98-119) and the
// ejection-or-close-encounter composite (composites.hpp:145-222, stop_on_close_encounter.hpp:101-142) -- without
// printing up to nbod 10, where launches with verbose_on_* keys go to the lane-per-system kernels, and with the
// reference's lprintf lines beyond.  The N lanes of a system act as one monitor object: every lane tests its own planet
// (and, for close encounters, the pairs (b, d), 1 <= d < b), ballots share the verdicts, every lane keeps the object's state
// (next_log_time, the close-encounter flag) and the lanes write a snapshot record together (log_system_body).
// The tests are the scalar helpers the lane-per-system monitors use (ej_test_quiet, ce_test_quiet), so outcomes,
// stop steps and records are the lane-per-system kernels' (tests/test_gpu_body_logging.py).
#pragma once
#include "device_common.hpp"

namespace swh {

enum { MONB_NONE = -1, MONB_OFF = 0, MONB_EJECTION = 1, MONB_LOG_INTERVAL = 2, MONB_EJECTION_OR_CLOSE = 3 };

// which of them a launch needs; MONB_NONE: a monitor only the lane-per-system kernels have (mce_stat, log_transit and,
// up to nbod 10, the printing variants: beyond, where no lane-per-system kernel exists, BodyMonitor prints itself)
inline int body_monitor_of(const MonParams &mp, const int nbod) {
    const int variant = monitor_variant(mp);
    if (variant == MON_OFF) return MONB_OFF;
    if (variant != MON_ON && !(variant == MON_VERBOSE && nbod > SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM)) return MONB_NONE;
    switch (mp.kind) {
    case SWARM_HIP_MON_EJECTION: return MONB_EJECTION;
    case SWARM_HIP_MON_LOG_INTERVAL: return MONB_LOG_INTERVAL;
    case SWARM_HIP_MON_EJECTION_OR_CLOSE: return MONB_EJECTION_OR_CLOSE;
    default: return MONB_NONE;
    }
}

// the lines of verbose_on_ejection / verbose_on_close_encounter (stop_on_ejection.hpp:120-130,
// stop_on_close_encounter.hpp:137-141): the operations and texts of ej_test_body / ce_check, written in line
// (log_printf_inline, device_common.hpp: a call would cost the kernels 4 - 9 % even where nothing is printed)
__device__ __forceinline__ void body_ejection_line(const LogBuf &lg, const int number, const int b, const double t,
                                                   const double (&p)[3], const double (&v)[3], const double m) {
#pragma clang fp contract(off)
    const double r = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const double speed_sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double epp = 0.5 * speed_sq * r / m - 1.;
    const double energy = 0.5 * speed_sq - m / r;
    if (fabs(epp) < 1e-4)
        SWH_LPRINTF_INLINE(lg, "Orbit is nearly parabolic: _sys=%d, bod=%d, T=%lg r=%lg energy=%lg energy*r/GM=%lg.\n",
                           parg(number), parg(b), parg(t), parg(r), parg(energy), parg(epp));
    else
        SWH_LPRINTF_INLINE(lg, "Orbit is hyperbolic: _sys=%d, bod=%d, T=%lg r=%lg energy=%lg energy*r/GM=%lg.\n",
                           parg(number), parg(b), parg(t), parg(r), parg(energy), parg(epp));
}
__device__ __forceinline__ void body_close_line(const LogBuf &lg, const int number, const double t, const int j,
                                                const int i, const double d) {
    SWH_LPRINTF_INLINE(lg, "Close apporach detected: sys=%d, T=%f j=%d i=%d  d=%lg.\n", parg(number), parg(t), parg(j),
                       parg(i), parg(d));
}

template <int N, int MONB, bool ROLLED = BodyLoops<N>::ROLLED>
struct BodyMonitor {
    // verbose_on_*: up to nbod 10 the lane-per-system kernels serve those launches (body_monitor_of)
    static constexpr bool PRINTS = N > SWARM_HIP_MAX_NBOD_LANE_PER_SYSTEM;

    double next_log_time; // log_time_interval.hpp:118-119
    bool ce_met;          // stop_on_close_encounter's condition_met, kept between calls as the monitor object does

    __device__ __forceinline__ void init(const double t) {
        next_log_time = t;
        ce_met = false;
    }

    // Monitor::operator() for the system whose body b this lane holds; called by the whole wavefront.
    //   take_part     the system really calls its monitor now (the same in all lanes of a system)
    //   star          lane of the system's body 0; group_mask: the lanes of the system
    //   mass(d)       mass of body d of my system, asked for d < b only
    //   number        index of the system in the ensemble (the _sys= of the printed lines)
    //   convert(who, to_std)   for a propagator whose state is not in standard coordinates: the systems `who` go to
    //                 standard coordinates for their record and back (generic_gpu_bppt_integrator.hpp:189-224)
    template <class Mass, class Convert>
    __device__ __forceinline__ void operator()(const bool take_part, const int b, const int star,
                                               const unsigned long long group_mask, const double (&p)[3],
                                               const double (&v)[3], const double m_own, Mass &&mass, const double t,
                                               const int id, const int number, int &state, const MonParams &mp,
                                               const LogBuf &lg, Convert &&convert) {
        if (MONB == MONB_OFF) return;
        [[maybe_unused]] constexpr int BODY_TRIP = BodyLoops<N>::trip(ROLLED);
        bool want = false, ej_met = false;
        if (MONB == MONB_EJECTION || MONB == MONB_EJECTION_OR_CLOSE) { // pass_one, stop_on_ejection.hpp:177-193
            const bool ej = take_part && b >= 1 && mp.ej_any && ej_test_quiet(p, v, m_own, mp);
            const unsigned long long hits = __ballot(ej) & group_mask;
            ej_met = hits != 0;
            if constexpr (PRINTS) { // the planets are tested in index order until one is met (L184-185): it alone prints
                if (mp.ej_verbose && ej && star + b == __ffsll((long long)hits) - 1)
                    body_ejection_line(lg, number, b, t, p, v, m_own);
            }
            want = take_part && ej_met && mp.ej_log;
        }
        if (MONB == MONB_EJECTION_OR_CLOSE) {
            // skipped for a system whose ejection is about to be logged: the composite's short-circuit ||
            const bool ce_part = take_part && !want;
            bool ce = false;
            [[maybe_unused]] int first_d = 0; // PRINTS: the first d < b that is close to b, 0 = none: stands for `ce`
            if (mp.ce_any) {
                SWH_BODY_LOOP
                for (int d = 1; d < N - 1; d++) {
                    const double pj[3] = {__shfl(p[0], star + d), __shfl(p[1], star + d), __shfl(p[2], star + d)};
                    double sep;
                    const bool c = ce_test_quiet(p, pj, m_own, mass(d), mass(0), mp, sep);
                    if constexpr (PRINTS)
                        first_d = (first_d == 0 && d < b && c) ? d : first_d;
                    else
                        ce = ce || (d < b && c);
                }
                if constexpr (PRINTS) ce = first_d != 0;
            }
            const unsigned long long ce_hits = __ballot(ce_part && ce) & group_mask;
            const bool ce_now = ce_hits != 0;
            // the pairs (b, d < b) are tested in order until one is met (stop_on_close_encounter.hpp:108-110): it alone
            // prints; its separation is formed again here, the whole wavefront together (the shuffles need every lane)
            if constexpr (PRINTS) {
                const bool printer = mp.ce_verbose && ce_part && ce && star + b == __ffsll((long long)ce_hits) - 1;
                if (__any(printer)) {
                    const double pj[3] = {__shfl(p[0], star + first_d), __shfl(p[1], star + first_d),
                                          __shfl(p[2], star + first_d)};
                    if (printer) {
#pragma clang fp contract(off)
                        const double dx = p[0] - pj[0], dy = p[1] - pj[1], dz = p[2] - pj[2];
                        body_close_line(lg, number, t, first_d, b, sqrt(dx * dx + dy * dy + dz * dz));
                    }
                }
            }
            if (ce_part) ce_met = ce_now;
            want = take_part && ((mp.ej_log && ej_met) || (mp.ce_log && ce_met));
        }
        if (MONB == MONB_LOG_INTERVAL) { // log_time_interval.hpp:98-111
#pragma clang fp contract(off)
            want = take_part && mp.log_interval != 0. && t >= next_log_time;
            if (want) next_log_time += mp.log_interval;
        }
        const bool logs = MONB == MONB_LOG_INTERVAL || mp.ej_log || (MONB == MONB_EJECTION_OR_CLOSE && mp.ce_log);
        const bool any_rec = logs && __any(want);
        if (any_rec) convert(want, true);
        // pass_two: stop_on_ejection.hpp:196-201, stop_on_close_encounter.hpp:118-123
        if (MONB != MONB_LOG_INTERVAL) {
            if (take_part && ej_met && mp.ej_deactivate) state = SWARM_HIP_DISABLED;
            if (MONB == MONB_EJECTION_OR_CLOSE && take_part && ce_met && mp.ce_deactivate) state = SWARM_HIP_DISABLED;
        }
        if (any_rec) {
            log_system_body<N>(lg, want, b, star, t, id, state, p, v, m_own);
            convert(want, false);
        }
    }
};

} // namespace swh
