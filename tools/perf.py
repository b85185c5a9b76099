#!/usr/bin/env python3
"""Kernel-level timing of any integrator/config on one GPU (not the headline bench).
Prints one JSON line: accepted/attempted system-steps per second, kernel ms (HIP events)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import swarm_amd as S

WORKLOADS = {
    # BASELINE.json configs
    "c2": dict(integrator="hermite", nsys=65536, nbod=3, T=2.0, cfg={"time_step": "0.0001"}),
    "c3": dict(integrator="rkck_adaptive", nsys=65536, nbod=6, T=1.0,
               cfg={"min_time_step": "0.0001", "max_time_step": "0.001", "error_tolerance": "1e-27",
                    "deactivate_on_ejection": "1", "rmax": "1000"}),
    "c3_nomon": dict(integrator="rkck_adaptive", nsys=65536, nbod=6, T=1.0,
                     cfg={"min_time_step": "0.0001", "max_time_step": "0.001", "error_tolerance": "1e-27"}),
    "c3_ce": dict(integrator="rkck_adaptive_close_encounter", nsys=65536, nbod=6, T=1.0,
                  cfg={"min_time_step": "0.0001", "max_time_step": "0.001", "error_tolerance": "1e-27",
                       "deactivate_on_close_encounter": "1", "close_approach": "0.01"}),
    "c4": dict(integrator="mvs", nsys=131072, nbod=8, T=10.0, cfg={"time_step": "0.01"}),  # one 1/8 shard of 2^20
    "c4_mon": dict(integrator="mvs", nsys=131072, nbod=8, T=10.0,
                   cfg={"time_step": "0.01", "deactivate_on_ejection": "1", "rmax": "1000"}),
    # Hermite_Adaptive.cfg + log_time_interval; T=100 (~270 adaptive steps per system), one snapshot
    # per system every 5 time units => 21 records x 288 B x 65536 = 396 MB of log per run
    "c5": dict(integrator="hermite_adap_log", nsys=65536, nbod=4, T=100.0,
               cfg={"time_step_factor": "170e-4", "min_time_step": "0.00001", "log_interval": "5"}, log=420 << 20),
    # Hermite.cfg with the ejection monitor armed (MON_ON variant of the kernel)
    "c2_mon": dict(integrator="hermite", nsys=65536, nbod=3, T=1.0,
                   cfg={"time_step": "0.0001", "deactivate_on_ejection": "1", "rmax": "1000"}),
    # the close-encounter composite on the fixed-step kernel: evaluated every step
    "c2_ce": dict(integrator="hermite_cpu_ejection_or_close_encounter", nsys=65536, nbod=3, T=1.0,
                  cfg={"time_step": "0.0001", "deactivate_on_close_encounter": "1", "close_approach": "0.01",
                       "deactivate_on_ejection": "1", "rmax": "1000"}),
    # Gauss IRK, test/integrators/irk2.cfg step
    "irk2": dict(integrator="irk2", nsys=65536, nbod=3, T=10.0, cfg={"time_step": "0.01"}),
    # the tutorial propagators at the time step of the reference's test/integrators/*.cfg
    "euler": dict(integrator="euler", nsys=65536, nbod=3, T=0.5, cfg={"time_step": "0.001"}),
    "midpoint": dict(integrator="midpoint", nsys=65536, nbod=3, T=0.5, cfg={"time_step": "0.001"}),
    "hermite_prop": dict(integrator="hermite_prop", nsys=65536, nbod=3, T=0.5, cfg={"time_step": "0.001"}),
    "verlet": dict(integrator="verlet", nsys=65536, nbod=3, T=0.5, cfg={"max_timestep": "0.001", "timestep_scale": "1.0"}),
    "c5_nolog": dict(integrator="hermite_adap", nsys=65536, nbod=4, T=100.0,
                     cfg={"time_step_factor": "170e-4", "min_time_step": "0.00001"}),
    # C3 where it is divergent: geometrically scaled systems (size factor 0.03 .. 1, dynamical times spread over a
    # factor ~190), a tolerance that rejects attempts, rmax that ejects the widest ~8 % at once; --shuffle puts the
    # systems in random order (every wave mixes fast and slow systems), without it neighbours are alike
    "c3_div": dict(integrator="rkck_adaptive", nsys=65536, nbod=6, T=2.0, ensemble="scaled",
                   cfg={"min_time_step": "1e-5", "max_time_step": "0.02", "error_tolerance": "1e-24",
                        "deactivate_on_ejection": "1", "rmax": "30"}),
}


def hermite_flops(n, adaptive=False):   # 118 P + 159 n (+ 9 n + 4 for the adaptive step size)
    return 118 * (n * (n - 1) // 2) + 159 * n + ((9 * n + 4) if adaptive else 0)


def rkck_flops(n):                      # per ATTEMPTED step: 150 P + 390 n
    return 150 * (n * (n - 1) // 2) + 390 * n


# mean Laguerre passes per Kepler solve on the generated ensemble at dt = 0.01 (counted with the CPU restatement:
# every solve makes 3 passes, 26 % a fourth, 10 % all seven)
MVS_MEAN_PASSES = 3.56


def mvs_flops(n, passes=MVS_MEAN_PASSES):   # 25 (n-1)(n-2)/2 + (n-1)(50 iters + 70) + 36 n
    return 25 * ((n - 1) * (n - 2) // 2) + (n - 1) * (50 * passes + 70) + 36 * n


def flops_per_step(integrator, nbod):
    """algorithmic flops per accepted step -- per ATTEMPTED step for rkck; None where the survey has none"""
    if integrator.startswith("hermite_adap"):
        return hermite_flops(nbod, True)
    if integrator.startswith("hermite"):
        return hermite_flops(nbod)
    if integrator.startswith("rkck"):
        return rkck_flops(nbod)
    if integrator.startswith("mvs"):
        return mvs_flops(nbod)
    return None


FP64_VECTOR_PEAK_TFLOPS = 78.6


def make_ensemble(w, nsys, nbod, spread=0.0, shuffle=False):
    host = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
    if w.get("ensemble") == "scaled":      # tests/conftest.py `scaled`: perturbed, then sized 0.03 .. 1
        rng = np.random.default_rng(3)
        host.pos[:, 1:, :] *= 1.0 + 1e-2 * rng.standard_normal((nsys, nbod - 1, 3))
        host.vel[:, 1:, :] *= 1.0 + 1e-2 * rng.standard_normal((nsys, nbod - 1, 3))
        host.pos[:, 1:, 2] += 1e-2 * rng.standard_normal((nsys, nbod - 1))
        s = np.geomspace(0.03, 1.0, nsys)
        if shuffle:
            rng.shuffle(s)
        s = s[:, None, None]
        host.pos *= s
        host.vel /= np.sqrt(s)
    if spread > 1:
        s = np.geomspace(1.0 / spread, 1.0, nsys)[:, None, None]
        host.pos *= s
        host.vel /= np.sqrt(s)
    return host


def lane_utilisation(att):
    """sum of lane-attempts / (64 x wave-attempts): a wave runs until its slowest lane is done"""
    att = np.asarray(att, dtype=np.float64)
    pad = (-len(att)) % 64
    waves = np.concatenate([att, np.zeros(pad)]).reshape(-1, 64)
    return float(att.sum() / max(64.0 * waves.max(axis=1).sum(), 1.0))


def measure(workload, nsys=None, nbod=None, T=None, reps=3, block=0, lane_mode=0, spread=0.0, shuffle=False,
            pinned=False, host_to_host=True, extra_cfg=None, warm=False, sparse=0):
    """kernel time of a run's launches (HIP events on the launch stream), the median repetition of `reps` with min and
    max beside it; step counts; optional host-to-host time of integrate()"""
    w = dict(WORKLOADS[workload])
    nsys, nbod, T = nsys or w["nsys"], nbod or w["nbod"], T or w["T"]
    S.capi.check(S.capi.lib.swarm_hip_init(0))
    L = S.capi.lib
    host0 = make_ensemble(w, nsys, nbod, spread, shuffle)
    if sparse > 1:                         # every `sparse`-th system ACTIVE, the others DISABLED: thin wavefronts
        host0.state[:] = S.capi.DISABLED
        host0.state[::sparse] = S.capi.ACTIVE
    elif sparse < 0:                       # one thin wavefront (4 ACTIVE systems) in every |sparse|, the others full
        st = host0.state.reshape(-1, 64) if nsys % 64 == 0 else None
        thin = np.arange(nsys // 64) % (-sparse) == 0
        st[thin, :] = S.capi.DISABLED
        st[thin, ::16] = S.capi.ACTIVE
    cfg = dict(w["cfg"], integrator=w["integrator"], lane_mode=str(lane_mode))
    cfg.update(extra_cfg or {})
    res = []
    for rep in range(reps):
        host = host0.clone()
        integ = S.Integrator.create(S.config(cfg))
        integ.set_ensemble(host)
        integ.set_destination_time(T)
        if block:
            integ.args.block_size = block
        lm = None
        if w.get("log"):
            lm = S.LogManager(w["log"], "count")
            integ.set_log_manager(lm)
        integ.enable_step_counters()
        integ.upload_ensemble()
        e0 = C.c_void_p(); e1 = C.c_void_p()
        L.swarm_hip_event_create(C.byref(e0)); L.swarm_hip_event_create(C.byref(e1))
        if warm:
            # the same launch on a copy of the ensemble, enqueued directly ahead of the timed one: a launch that
            # follows an idle gap runs its first milliseconds at a lower clock (C4 shard: 10.65 ms alone, 9.71 ms
            # back to back, tools/exp_b2b.py), and the secondary configurations are 0.6 .. 11 ms long
            wh = host0.clone()
            wi = S.Integrator.create(S.config(cfg))
            wi.set_ensemble(wh)
            wi.set_destination_time(T)
            if block:
                wi.args.block_size = block
            if w.get("log"):
                wi.set_log_manager(lm)      # same buffer: flushed (and its cursor cleared) before the timed launch's flush
            wi.upload_ensemble()
            wi.launch_integrator()
            if not lm:                      # and once more over the next span of time: the ramp outlasts a 10 ms launch
                wi.set_destination_time(2 * T)
                wi._dens.reactivate(None)
                wi.launch_integrator()
            if lm:
                L.swarm_hip_synchronize(None)
                lm.flush()
                lm.bytes_flushed = 0
                wi.launch_integrator()      # (already finished systems: returns at once; keeps the queue non-empty)
        L.swarm_hip_event_record(e0, None)
        t0 = time.perf_counter()
        integ.launch_integrator()
        L.swarm_hip_event_record(e1, None)
        L.swarm_hip_synchronize(None)
        tk = time.perf_counter()
        integ.flush_log()
        tf = time.perf_counter()
        ms = C.c_float()
        L.swarm_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        acc1, att1 = integ.step_counts()         # of the timed launch alone
        attempts = 1
        # the launches that follow (rkck caps a launch at destination_time / max_time_step attempts,
        # rkck_integrator.hpp:75): lane-attempts and wave-attempts of each, for the utilisation over the whole run
        lane_att, wave_att, prev = float(att1.sum()), 64.0 * np.pad(att1.astype(np.float64), (0, (-len(att1)) % 64)).reshape(-1, 64).max(axis=1).sum(), att1
        launch_ms = [ms.value]
        # per launch: ACTIVE systems when it started, attempts made in it (sum, and of its slowest system), and
        # sum / (64 x sum over index-order wavefronts of the wave's slowest): the lane utilisation of that launch if it
        # runs one lane per system without gathering
        launches = [dict(active=int((host0.state == S.capi.ACTIVE).sum()), ms=round(ms.value, 4), attempts=int(att1.sum()),
                         slowest=int(att1.max()), lane_utilisation=round(lane_utilisation(att1), 4))]
        while True:
            n_active = integ._dens.count_active()
            if n_active == 0 or attempts >= integ._max_attempts:
                break
            L.swarm_hip_event_record(e0, None)
            integ.launch_integrator()
            L.swarm_hip_event_record(e1, None)
            L.swarm_hip_synchronize(None)
            integ.flush_log()
            m2 = C.c_float()
            L.swarm_hip_event_elapsed_ms(e0, e1, C.byref(m2))
            launch_ms.append(m2.value)
            attempts += 1
            _, now = integ.step_counts()
            d = (now - prev).astype(np.float64)
            prev = now
            lane_att += d.sum()
            wave_att += 64.0 * np.pad(d, (0, (-len(d)) % 64)).reshape(-1, 64).max(axis=1).sum()
            launches.append(dict(active=int(n_active), ms=round(m2.value, 4), attempts=int(d.sum()), slowest=int(d.max()),
                                 lane_utilisation=round(lane_utilisation(d), 4)))
        wall = time.perf_counter() - t0
        acc, att = integ.step_counts()
        integ.download_ensemble()
        L.swarm_hip_event_destroy(e0); L.swarm_hip_event_destroy(e1)
        res.append(dict(ms=ms.value, wall_ms=wall * 1e3, accepted=int(acc1.sum()), attempted=int(att1.sum()),
                        accepted_all_launches=int(acc.sum()), attempts=attempts, flush_ms=(tf - tk) * 1e3,
                        log_bytes=lm.bytes_flushed if lm else 0, dropped=lm.dropped if lm else 0,
                        lane_utilisation=lane_utilisation(att1), lane_utilisation_all_launches=lane_att / max(wave_att, 1.0),
                        attempted_all_launches=int(att.sum()), kernel_ms_all_launches=sum(launch_ms),
                        launch_ms=[round(x, 4) for x in launch_ms], launches=launches,
                        accepted_min_max=[int(acc.min()), int(acc.max())],
                        states={int(k): int(v) for k, v in zip(*np.unique(host.state, return_counts=True))}))
    # the reported repetition is the MEDIAN one by kernel time over all launches of the run; min and max beside it
    # (the first repetition of a process carries the code-object load of every kernel it is the first to use)
    order = sorted(res, key=lambda r: r["kernel_ms_all_launches"])
    best = dict(order[len(order) // 2])
    best["reps"] = len(res)
    best["ms_min_max"] = [round(min(r["ms"] for r in res), 4), round(max(r["ms"] for r in res), 4)]
    best["kernel_ms_all_launches_min_max"] = [round(order[0]["kernel_ms_all_launches"], 4),
                                              round(order[-1]["kernel_ms_all_launches"], 4)]
    if host_to_host:
        # host-to-host time of integrator.integrate(): upload (pack + H2D), kernel(s), flush, D2H + unpack
        e2e_ms = float("inf")
        for _ in range(2):  # best of two: the host work above leaves the GPU idle long enough for its clock to drop
            host = host0.clone(pinned=pinned)
            integ = S.Integrator.create(S.config(cfg))
            integ.set_ensemble(host)
            integ.set_destination_time(T)
            if w.get("log"):
                integ.set_log_manager(S.LogManager(w["log"], "count"))
            t0 = time.perf_counter()
            integ.integrate()
            e2e_ms = min(e2e_ms, (time.perf_counter() - t0) * 1e3)
        best["host_to_host_ms"] = e2e_ms
        best["pinned_host"] = bool(pinned)
    out = dict(workload=workload, integrator=w["integrator"], nsys=nsys, nbod=nbod, T=T, **best)
    out["accepted_steps_per_s"] = best["accepted"] / (best["ms"] * 1e-3)
    out["attempted_steps_per_s"] = best["attempted"] / (best["ms"] * 1e-3)
    f = flops_per_step(w["integrator"], nbod)
    if f:
        out["flop_per_step"] = f
        out["algorithmic_tflops"] = f * best["attempted"] / (best["ms"] * 1e-3) / 1e12
        out["frac_of_fp64_peak"] = out["algorithmic_tflops"] / FP64_VECTOR_PEAK_TFLOPS
    return out




def measure_flush(workload, batches=0, nsys=None, nbod=None, T=None, reps=5, capacity=None):
    """Host-to-host time of the launch loop WITH its log flushes (core_integrate: launches, flush after each, active
    count; ensemble resident, records into page-locked memory, no sink): batches = 0 is the reference's protocol (one
    log, a synchronous copy after the launch), batches >= 1 the two-log manager with every attempt split into that
    many launches over runs of wave chunks.  The same LogManager serves all repetitions and the first
    one is dropped (page-locked memory is slow the first time it is written).  Median of the rest."""
    w = dict(WORKLOADS[workload])
    nsys, nbod, T = nsys or w["nsys"], nbod or w["nbod"], T or w["T"]
    S.capi.check(S.capi.lib.swarm_hip_init(0))
    L = S.capi.lib
    host0 = make_ensemble(w, nsys, nbod)
    cfg = dict(w["cfg"], integrator=w["integrator"], log_batches=str(max(batches, 1)))
    cap = capacity or (w["log"] if batches == 0 else (w["log"] // batches + (8 << 20)))
    lm = S.LogManager(cap, "count", buffers=2 if batches else 1)
    res = []
    for rep in range(reps + 1):
        host = host0.clone()
        integ = S.Integrator.create(S.config(cfg))
        integ.set_ensemble(host)
        integ.set_destination_time(T)
        integ.set_log_manager(lm)
        integ.upload_ensemble()
        lm.bytes_flushed = 0
        lm.dropped = 0
        L.swarm_hip_synchronize(None)
        t0 = time.perf_counter()
        attempts = integ.core_integrate()
        L.swarm_hip_synchronize(None)
        ms = (time.perf_counter() - t0) * 1e3
        if rep:
            res.append(dict(ms=ms, log_bytes=lm.bytes_flushed, dropped=lm.dropped, attempts=attempts))
    res.sort(key=lambda r: r["ms"])
    out = dict(res[len(res) // 2])
    out.update(workload=workload, nsys=nsys, nbod=nbod, T=T, batches=batches, log_capacity=cap, reps=reps,
               ms_min_max=[round(res[0]["ms"], 4), round(res[-1]["ms"], 4)],
               d2h_GBps=out["log_bytes"] / (out["ms"] * 1e-3) / 1e9)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(WORKLOADS))
    ap.add_argument("--nsys", type=int)
    ap.add_argument("--nbod", type=int)
    ap.add_argument("--T", type=float)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--lane-mode", type=int, default=0, help="0 auto, 1 lane per system, 2 lane per body (hermite, hermite_adap, mvs, irk2, rkck)")
    ap.add_argument("--spread", type=float, default=0.0, help="scale systems geometrically by up to this factor")
    ap.add_argument("--shuffle", action="store_true", help="c3_div: systems in random order")
    ap.add_argument("--pinned", action="store_true", help="host ensemble in page-locked memory (swarm::hostEnsemble)")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--warm", action="store_true", help="enqueue the same launch on a copy directly ahead of the timed one")
    ap.add_argument("--sparse", type=int, default=0, help="K > 1: only every K-th system ACTIVE, the others DISABLED; K < 0: every |K|-th wavefront thin (4 ACTIVE), the others full")
    ap.add_argument("--flush-batches", type=int, default=None,
                    help="logging workloads: time the launch loop with its flushes instead (0 = one log, synchronous "
                         "copy; B >= 1 = two logs, B launches per attempt, copies overlapping launches)")
    ap.add_argument("cfg", nargs="*", help="extra key=value configuration pairs")
    a = ap.parse_args()
    if a.flush_batches is not None:
        print(json.dumps(measure_flush(a.workload, a.flush_batches, a.nsys, a.nbod, a.T, a.reps)))
        return
    print(json.dumps(measure(a.workload, a.nsys, a.nbod, a.T, a.reps, a.block, a.lane_mode, a.spread, a.shuffle,
                             a.pinned, not a.no_host_to_host, dict(kv.split("=", 1) for kv in a.cfg), a.warm, a.sparse)))


if __name__ == "__main__":
    main()
