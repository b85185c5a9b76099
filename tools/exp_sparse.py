#!/usr/bin/env python3
"""experiment: kernel time vs fraction of ACTIVE lanes per wave (the rest DISABLED from the start)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swarm_amd as S

def run(integrator, cfg, nbod, T, every, blocked=False):
    nsys = 65536
    host = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
    host.state[:] = -1
    if blocked:
        host.state[: nsys // every] = 0
    else:
        host.state[::every] = 0
    integ = S.Integrator.create(S.config(dict(cfg, integrator=integrator))); integ.set_ensemble(host); integ.set_destination_time(T)
    integ.upload_ensemble()
    L = S.capi.lib
    e0 = C.c_void_p(); e1 = C.c_void_p(); L.swarm_hip_event_create(C.byref(e0)); L.swarm_hip_event_create(C.byref(e1))
    best = 1e9
    for k in range(3):
        integ._dens.upload(host)
        L.swarm_hip_event_record(e0, None); integ.launch_integrator(); L.swarm_hip_event_record(e1, None); L.swarm_hip_synchronize(None)
        ms = C.c_float(); L.swarm_hip_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value)
    return round(best, 3)

for name, cfg, nbod, T in (("hermite", {"time_step": "0.001"}, 3, 2.0), ("hermite", {"time_step": "0.001"}, 6, 1.0),
                           ("rkck_fixed", {"min_time_step": "1e-3", "max_time_step": "1e-3", "error_tolerance": "1e-27"}, 6, 0.2),
                           ("rkck_adaptive", {"min_time_step": "1e-3", "max_time_step": "1e-3", "error_tolerance": "1e-27"}, 6, 0.2),
                           ("rkck_adaptive", {"min_time_step": "1e-3", "max_time_step": "1e-3", "error_tolerance": "1e-27"}, 3, 0.2),
                           ("mvs", {"time_step": "0.01"}, 8, 2.0)):
    print(name, nbod, {("every %d" % e): run(name, cfg, nbod, T, e) for e in (1, 2, 10, 64)}, "first tenth:", run(name, cfg, nbod, T, 10, True), flush=True)
