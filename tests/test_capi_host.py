"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the
host mirror of the reference interface behaves like the reference (config, registry, errors)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from swarm_amd import capi
    hdr = open(os.path.join(ROOT, "include", "swarm_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|int64_t|const char \*)\s*(swarm_hip_[a-z0-9_]+)\(", hdr, re.M))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(capi.lib, s), s
    assert b"gfx950" in capi.lib.swarm_hip_version()


def test_layout_helpers():
    from swarm_amd import capi
    L = capi.lib
    assert L.swarm_hip_nchunk(0) == 0 and L.swarm_hip_nchunk(1) == 1 and L.swarm_hip_nchunk(65) == 2
    assert L.swarm_hip_body_bytes(65536, 3, 1) == 65536 * 3 * 8 * 8
    assert L.swarm_hip_sysd_bytes(64, 1) == 64 * 2 * 8
    assert L.swarm_hip_sysi_bytes(100) == 128 * 2 * 4
    assert L.swarm_hip_snapshot_record_bytes(3) == 224 and L.swarm_hip_snapshot_record_bytes(4) == 288


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (checked against a gcc-compiled probe)"""
    import ctypes, subprocess, tempfile
    from swarm_amd import capi
    src = '#include <stdio.h>\n#include "swarm_hip.h"\nint main(){printf("%zu %zu %zu\\n",' \
          'sizeof(swarm_hip_ensemble_t),sizeof(swarm_hip_log_t),sizeof(swarm_hip_launch_t));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o",
                               os.path.join(d, "p")])
        out = subprocess.check_output([os.path.join(d, "p")]).split()
    assert [int(x) for x in out] == [ctypes.sizeof(capi.EnsembleStruct), ctypes.sizeof(capi.LogStruct),
                                     ctypes.sizeof(capi.LaunchStruct)]


def test_argument_errors_without_gpu():
    """bad shapes are rejected before any HIP call (return code + message, never a throw)"""
    import ctypes as C
    from swarm_amd import capi
    e = capi.EnsembleStruct(16, 2, 1, 1, 8, 8, 8)
    a = capi.LaunchStruct()
    assert capi.lib.swarm_hip_launch_integrator(C.byref(e), C.byref(a)) == -1
    assert b"nbod" in capi.lib.swarm_hip_last_error()
    e = capi.EnsembleStruct(16, 3, 1, 1, None, None, None)
    assert capi.lib.swarm_hip_launch_integrator(C.byref(e), C.byref(a)) == -1
    with pytest.raises(capi.SwarmHipError):
        capi.check(capi.lib.swarm_hip_reactivate(C.byref(e), None))
    e = capi.EnsembleStruct(16, capi.MAX_NBOD + 1, 1, 1, 8, 8, 8)
    assert capi.lib.swarm_hip_launch_integrator(C.byref(e), C.byref(a)) == -1
    e = capi.EnsembleStruct(16, 3, 1, 1, 8, 12, 8)             # sysd not 8-byte aligned
    assert capi.lib.swarm_hip_launch_integrator(C.byref(e), C.byref(a)) == -1
    assert b"aligned" in capi.lib.swarm_hip_last_error()
    e = capi.EnsembleStruct((1 << 30) + 1, 3, 1, 1, 8, 8, 8)
    assert capi.lib.swarm_hip_launch_integrator(C.byref(e), C.byref(a)) == -1
    assert capi.lib.swarm_hip_launch_integrator(None, C.byref(a)) == -1
    assert capi.lib.swarm_hip_memcpy_h2d(None, None, 16) == -1
    assert capi.lib.swarm_hip_event_create(None) == -1


def test_config_semantics(tmp_path):
    import swarm_amd as S
    p = tmp_path / "x.cfg"
    p.write_text("# comment\n\nintegrator=hermite\ntime_step = 0.001\ntime_step=0.002\nflag=true\nn=12abc\n")
    cfg = S.config.load(str(p))
    assert cfg["integrator"] == "hermite" and cfg.require("time_step", 0.0) == 0.002
    assert cfg.optional("flag", False) is False          # `=true` silently yields the default
    assert cfg.optional("n", 0) == 12                     # sscanf %i stops at the first bad char
    assert cfg.optional("missing", 1.5) == 1.5
    with pytest.raises(S.key_not_found):
        cfg.require("destination_time", 0.0)
    assert S.default_config()["integrator"] == "hermite_cpu"


def test_plugin_registry_names_and_errors():
    import swarm_amd as S
    for name in ("hermite", "hermite_adap", "hermite_adap_log", "rkck_fixed", "rkck_adaptive", "mvs", "verlet",
                 "euler", "midpoint", "hermite_adap_close_encounter", "mvs_close_encounter", "hermite_prop"):
        assert "integrator_" + name in S.PLUGINS
    with pytest.raises(S.runtime_error, match="Integrator hermite_log not found"):
        S.Integrator.create(S.config({"integrator": "hermite_log"}))      # commented out in the fork
    with pytest.raises(S.runtime_error, match="has to be chosen"):
        S.Integrator.create(S.config({}))
    with pytest.raises(S.key_not_found):
        S.Integrator.create(S.config({"integrator": "hermite"}))         # time_step is required
    with pytest.raises(S.runtime_error, match="rkck requires"):
        S.Integrator.create(S.config({"integrator": "rkck_adaptive", "min_time_step": "1e-4"}))
    with pytest.raises(S.key_not_found):
        S.Integrator.create(S.config({"integrator": "hermite_adap_log", "time_step_factor": "0.02",
                                      "min_time_step": "1e-7"}))         # log_interval is required


def test_generate_ensemble_matches_oracle(oracle):
    import swarm_amd as S
    for nbod in (3, 6, 8):
        cfg = S.config({"nsys": "7", "nbod": str(nbod), "spacing_factor": "1.3"})
        g = S.generate_ensemble(cfg)
        o = oracle.generate(7, nbod, spacing_factor=1.3)
        assert (g.pos == o.pos).all() and (g.vel == o.vel).all() and (g.mass == o.mass).all()
        assert (g.body_attr == o.body_attr).all() and (g.state == 0).all()
        np.testing.assert_allclose(g.calc_total_energy(), oracle.total_energy(o), rtol=1e-14)


def test_printf_record_layout_and_replay(oracle):
    """MSG_PRINTF records (gpulog_printf.h, gpulog_msg_layout.h:112-204): header | char fmt[N] at 8 |
    arguments at natural alignment | pad 16; run_printf (lprintf.h:74-171) formats them.  The oracle
    writes one through its mce_stat restatement; swarm_amd.gpulog (product host code) replays it."""
    import io
    import struct
    from swarm_amd import gpulog
    O = oracle
    e = O.Ensemble(1, 3, 3, 1)
    g = O.generate(1, 3, planet_mass=1e-3)
    for k in ("pos", "vel", "mass", "time", "state", "id"):
        setattr(e, k, getattr(g, k).copy())
    e.body_attr[:, 0, 0], e.body_attr[:, 1:, 0] = 0.1, 0.05
    e.id[0] = 7
    e.pos[0, 1], e.vel[0, 1] = [10, 0, 0], [0, np.sqrt(.1), 0]
    e.pos[0, 2], e.vel[0, 2] = [10 * np.cos(.05), 10 * np.sin(.05), 0], [np.sqrt(.1) * np.sin(.05), -np.sqrt(.1) * np.cos(.05), 0]
    log = O.Log(1 << 16)
    O.integrate(e, O.params("hermite", 2.0, time_step=0.01), O.monitor(kind=O.MON_MCE_STAT, mce_time_step=0.01), log)
    raw = log.bytes()
    recs = list(gpulog.records(raw))
    assert len(recs) == 1 and recs[0][0] == gpulog.MSG_PRINTF
    rec = recs[0][1]
    fmt = b"Hitting detected: sysid = %d, i = %d, j = %d, time= %f\n\0"
    assert rec[8:8 + len(fmt)] == fmt
    at = (8 + len(fmt) + 3) // 4 * 4
    assert struct.unpack_from("<iii", rec, at) == (7, 1, 2)
    at = (at + 12 + 7) // 8 * 8
    t = struct.unpack_from("<d", rec, at)[0]
    assert len(rec) == (at + 8 + 15) // 16 * 16 == struct.unpack_from("<i", rec, 4)[0]
    assert gpulog.run_printf(rec) == "Hitting detected: sysid = 7, i = 1, j = 2, time= %f\n" % t
    out = io.StringIO()
    assert gpulog.replay_printf(out, raw) == 1 and out.getvalue() == gpulog.run_printf(rec)
    assert e.mass[0, 2] == 0 and e.mass[0, 1] == np.float32(2e-3)


def test_oracle_mce_swapped_loop_variables_halt_the_system(oracle):
    """mercury_mce_stat.hpp:312 swaps the loop variables i,j when the lower-index planet is lighter;
    the pair is then visited again with one mass already zero -> "3 planets in collision" + DISABLED."""
    O = oracle
    e = O.Ensemble(1, 3, 3, 1)
    g = O.generate(1, 3, planet_mass=1e-3)
    for k in ("pos", "vel", "mass", "time", "state", "id"):
        setattr(e, k, getattr(g, k).copy())
    e.body_attr[:, 0, 0], e.body_attr[:, 1:, 0] = 0.1, 0.05
    e.mass[0, 2] = 2e-3
    e.pos[0, 1], e.vel[0, 1] = [10, 0, 0], [0, np.sqrt(.1), 0]
    e.pos[0, 2], e.vel[0, 2] = [10 * np.cos(.05), 10 * np.sin(.05), 0], [np.sqrt(.1) * np.sin(.05), -np.sqrt(.1) * np.cos(.05), 0]
    log = O.Log(1 << 16)
    O.integrate(e, O.params("hermite", 2.0, time_step=0.01), O.monitor(kind=O.MON_MCE_STAT, mce_time_step=0.01), log)
    recs = log.records()
    assert [r["msgid"] for r in recs] == [-2, -2, 1]
    assert recs[0]["text"].startswith("Hitting detected: sysid = 0, i = 2, j = 1")
    assert recs[1]["text"].startswith("3 planets in collision: System halted ! T=")
    assert recs[2]["state"] == O.ACTIVE and e.state[0] == O.DISABLED   # logged before set_disabled (L345-346)
    assert e.mass[0, 1] == 0 and e.mass[0, 2] == np.float32(3e-3)
