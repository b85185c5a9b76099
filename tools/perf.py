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
    "c5_nolog": dict(integrator="hermite_adap", nsys=65536, nbod=4, T=100.0,
                     cfg={"time_step_factor": "170e-4", "min_time_step": "0.00001"}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=sorted(WORKLOADS))
    ap.add_argument("--nsys", type=int)
    ap.add_argument("--nbod", type=int)
    ap.add_argument("--T", type=float)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--lane-mode", type=int, default=0, help="0 auto, 1 lane per system, 2 lane per body (hermite)")
    ap.add_argument("--spread", type=float, default=0.0, help="scale systems geometrically by up to this factor")
    ap.add_argument("--pinned", action="store_true", help="host ensemble in page-locked memory (swarm::hostEnsemble)")
    a = ap.parse_args()
    w = dict(WORKLOADS[a.workload])
    nsys, nbod, T = a.nsys or w["nsys"], a.nbod or w["nbod"], a.T or w["T"]
    S.capi.check(S.capi.lib.swarm_hip_init(0))
    L = S.capi.lib
    host0 = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
    if a.spread > 1:
        s = np.geomspace(1.0 / a.spread, 1.0, nsys)[:, None, None]
        host0.pos *= s
        host0.vel /= np.sqrt(s)
    cfg = dict(w["cfg"], integrator=w["integrator"], lane_mode=str(a.lane_mode))
    res = []
    for rep in range(a.reps):
        host = host0.clone()
        integ = S.Integrator.create(S.config(cfg))
        integ.set_ensemble(host)
        integ.set_destination_time(T)
        if a.block:
            integ.args.block_size = a.block
        lm = None
        if w.get("log"):
            lm = S.LogManager(w["log"], "count")
            integ.set_log_manager(lm)
        integ.enable_step_counters()
        integ.upload_ensemble()
        e0 = C.c_void_p(); e1 = C.c_void_p()
        L.swarm_hip_event_create(C.byref(e0)); L.swarm_hip_event_create(C.byref(e1))
        L.swarm_hip_event_record(e0, None)
        t0 = time.perf_counter()
        integ.launch_integrator()
        L.swarm_hip_event_record(e1, None)
        L.swarm_hip_synchronize(None)
        tk = time.perf_counter()
        integ.flush_log()
        tf = time.perf_counter()
        attempts = 1
        if integ._dens.count_active() != 0:
            attempts += integ.core_integrate()
        ms = C.c_float()
        L.swarm_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        wall = time.perf_counter() - t0
        acc, att = integ.step_counts()
        integ.download_ensemble()
        res.append(dict(ms=ms.value, wall_ms=wall * 1e3, accepted=int(acc.sum()), attempted=int(att.sum()),
                        attempts=attempts, flush_ms=(tf - tk) * 1e3, log_bytes=lm.bytes_flushed if lm else 0, dropped=lm.dropped if lm else 0,
                        states={int(k): int(v) for k, v in zip(*np.unique(host.state, return_counts=True))}))
    # host-to-host time of integrator.integrate(): upload (pack + H2D), kernel(s), flush, D2H + unpack
    e2e_ms = float("inf")
    for _ in range(2):  # best of two: the host work above leaves the GPU idle long enough for its clock to drop
        host = host0.clone(pinned=a.pinned)
        integ = S.Integrator.create(S.config(cfg))
        integ.set_ensemble(host)
        integ.set_destination_time(T)
        if w.get("log"):
            integ.set_log_manager(S.LogManager(w["log"], "count"))
        t0 = time.perf_counter()
        integ.integrate()
        e2e_ms = min(e2e_ms, (time.perf_counter() - t0) * 1e3)
    best = min(res, key=lambda r: r["ms"])
    best["host_to_host_ms"] = e2e_ms
    best["pinned_host"] = bool(a.pinned)
    out = dict(workload=a.workload, integrator=w["integrator"], nsys=nsys, nbod=nbod, T=T, **best)
    out["accepted_steps_per_s"] = best["accepted"] / (best["ms"] * 1e-3)
    out["attempted_steps_per_s"] = best["attempted"] / (best["ms"] * 1e-3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
