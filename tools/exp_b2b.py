#!/usr/bin/env python3
"""experiment: the same launch enqueued back to back (no host gap): per-launch kernel time by HIP events.
  exp_b2b.py [nsys] [c4|c5] [launches]   (c4: mvs nbod 8, 1000 steps; c5: hermite_adap nbod 4, T = 100 per launch)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swarm_amd as S
wl = sys.argv[2] if len(sys.argv) > 2 else "c4"
nsys, nbod, n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 8 if wl == "c4" else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 8
span = 10.0 if wl == "c4" else 100.0
S.capi.check(S.capi.lib.swarm_hip_init(0))
L = S.capi.lib
host = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
integ = S.Integrator.create(S.config({"integrator": "mvs", "time_step": "0.01"} if wl == "c4" else
                                     {"integrator": "hermite_adap", "time_step_factor": "170e-4", "min_time_step": "0.00001"}))
integ.set_ensemble(host); integ.upload_ensemble()
ev = []
for k in range(n + 1):
    p = C.c_void_p(); L.swarm_hip_event_create(C.byref(p)); ev.append(p)
L.swarm_hip_event_record(ev[0], None)
for k in range(n):
    integ.set_destination_time(span * (k + 1))
    integ._dens.reactivate(None)
    integ.launch_integrator()
    L.swarm_hip_event_record(ev[k + 1], None)
L.swarm_hip_synchronize(None)
out = []
for k in range(n):
    ms = C.c_float(); L.swarm_hip_event_elapsed_ms(ev[k], ev[k + 1], C.byref(ms)); out.append(round(ms.value, 3))
print(nsys, out)
