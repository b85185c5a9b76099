#!/usr/bin/env python3
"""experiment: cost of a 100-attempt rkck launch vs. how the slow systems are spread over the waves"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swarm_amd as S
import perf

def run(tag, scale, extra=None):
    nsys, nbod = 65536, 6
    w = perf.WORKLOADS["c3_div"]
    host = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
    rng = np.random.default_rng(3)
    host.pos[:, 1:, :] *= 1.0 + 1e-2 * rng.standard_normal((nsys, nbod - 1, 3))
    s = scale[:, None, None]
    host.pos *= s; host.vel /= np.sqrt(s)
    cfg = dict(w["cfg"], integrator="rkck_adaptive"); cfg.update(extra or {})
    integ = S.Integrator.create(S.config(cfg)); integ.set_ensemble(host); integ.set_destination_time(2.0)
    integ.enable_step_counters(); integ.upload_ensemble()
    L = S.capi.lib
    e0 = C.c_void_p(); e1 = C.c_void_p(); L.swarm_hip_event_create(C.byref(e0)); L.swarm_hip_event_create(C.byref(e1))
    out = []
    prev = np.zeros(nsys)
    for k in range(12):
        L.swarm_hip_event_record(e0, None); integ.launch_integrator(); L.swarm_hip_event_record(e1, None); L.swarm_hip_synchronize(None)
        ms = C.c_float(); L.swarm_hip_event_elapsed_ms(e0, e1, C.byref(ms))
        _, att = integ.step_counts(); d = att - prev; prev = att.astype(np.float64)
        out.append((round(ms.value, 3), int(d.sum()), int((d > 0).sum())))
        if integ._dens.count_active() == 0: break
    print(tag, out)

n = 65536
small = np.full(n, 0.03)
run("all lanes scale 0.03", small)
mixed = np.ones(n); mixed[::10] = 0.03
run("every 10th lane 0.03, rest 1.0", mixed)
run("same, monitor off", mixed, {"deactivate_on_ejection": "0"})
blocks = np.ones(n); blocks[: n // 10] = 0.03
run("first tenth 0.03, rest 1.0", blocks)
