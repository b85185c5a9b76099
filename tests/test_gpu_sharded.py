"""The sharded harness with the product backend on a real GPU: swarm_amd.parallel.run_sharded + HipShardBackend
(what `bench.py --config c4` drives), at world = 1 in-process and at world = 2 through bench.py's own launcher
(two ranks sharing cuda:0, the exchange over gloo).  tests/test_parallel_gloo.py runs the same harness with a CPU
stand-in for the device at world 2 and 3."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, perturbed, rel_diff, to_host

pytestmark = pytest.mark.gpu


def _hip_shard_worker(port, q):
    """Runs in a fresh process, torch first: the PyTorch wheel and /opt/rocm each ship a libamdhip64.so.7, the first
    one loaded serves the whole process, and torch finds no GPU behind /opt/rocm's (bench.py imports torch first too)."""
    try:
        import torch
        torch.cuda.init()
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        import torch.distributed as dist
        import swarm_amd as S
        from oracle import oracle as O
        from swarm_amd import parallel as P
        O.build()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # RCCL with one rank
        S.capi.check(S.capi.lib.swarm_hip_init(0))
        nsys, nbod, T = 4000, 8, 2.0
        e = perturbed(O, nsys, nbod, seed=31, amp=2e-3)
        full0 = to_host(S, e)
        cfg = S.config({"integrator": "mvs", "time_step": "0.01", "kepler_compensated": "0"})
        res = P.run_sharded(nsys, lambda lo, hi: full0.slice(lo, hi), lambda cap, n: P.HipShardBackend(cap, n, nbod, dev),
                            cfg, T, dist=dist, rank=0, world=1, sync=torch.cuda.synchronize)
        assert res["capacity"] == 4032 and res["bounds"] == [(0, nsys)] and res["exchange"] == "gather to rank 0"
        assert res["checksums_match"] and res["ids_are_arange"]
        got = res["full"]
        # the unsharded run through the ordinary host API
        h = full0.clone()
        integ = S.Integrator.create(cfg)
        integ.set_ensemble(h)
        integ.set_destination_time(T)
        integ.integrate()
        for k in ("pos", "vel", "mass", "time", "state", "id"):
            assert np.array_equal(getattr(got, k), getattr(h, k)), k
        assert (got.state == O.INACTIVE).all() and (got.time == T).all()
        # statistics: the device reduction against utils.cpp:108-121 evaluated on the host ensembles
        err = np.abs((h.calc_total_energy() - full0.calc_total_energy()) / full0.calc_total_energy())
        st = res["stats"]
        assert st["nsys"] == nsys and st["n_inactive"] == nsys and st["n_active"] == 0 and st["n_disabled"] == 0
        # (the device sums K and U in the reference's order, numpy in its own: energies differ in the last bit, i.e.
        # the errors by ~1e-16 absolute)
        assert abs(st["max_abs_dE_over_E"] - err.max()) < 5e-15 and abs(st["min_abs_dE_over_E"] - err.min()) < 5e-15, (st, err.max())
        assert abs(st["sum_abs_dE_over_E"] - err.sum()) < 5e-15 * nsys and err.max() > 5e-14, (st, err.sum(), err.max())
        # a sample against the oracle at the C4 bars (tests/test_gpu_fullsize.py: pos 1e-10, vel 8e-10 per 1000 steps)
        idx = np.sort(np.random.default_rng(4).choice(nsys, 128, replace=False))
        o = O.Ensemble(len(idx), nbod)
        for k in ("pos", "vel", "mass", "time", "state", "id"):
            setattr(o, k, np.ascontiguousarray(getattr(e, k)[idx]))
        O.integrate(o, O.params("mvs", T, time_step=0.01), threads=8)
        assert (got.time[idx] == o.time).all() and (got.state[idx] == o.state).all()
        assert rel_diff(got.pos[idx, 1:], o.pos[:, 1:]) < 1e-10 and rel_diff(got.vel[idx, 1:], o.vel[:, 1:]) < 8e-10
        dist.destroy_process_group()
        q.put("ok")
    except BaseException:
        import traceback
        q.put(traceback.format_exc())


def test_run_sharded_with_hip_backend_is_the_unsharded_run(gpu):
    """mvs, 4000 x 8 (ragged: 63 chunks), one rank over RCCL: the image that run_sharded gathers is the unsharded
    integrate() bit for bit, ids are 0 .. n-1, the statistics vector of swarm_hip_energy_stats equals
    energy_conservation_error_range of the host result, and a sample agrees with the oracle at the C4 bars."""
    import socket
    import multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_hip_shard_worker, args=(port, q))
    p.start()
    out = q.get(timeout=280)
    p.join(30)
    assert out == "ok", out


def _bench(*argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no external launcher starts two ranks itself (here both on cuda:0, exchange
    over gloo: --rehearse-one-gpu) and rank 0 prints the one JSON line with n_gpus = 2, the aggregate of both shards
    and a verified gather; `--gpus 1` stays a single in-process rank."""
    two = _bench("--gpus", "2", "--rehearse-one-gpu", "--steps", "2", "--warmup", "1", "--span", "0.2",
                 "--no-cpu-baseline", "--no-secondary")
    assert two["n_gpus"] == 2 and two["gather_verified"] is True and two["rccl_ranks"] == 0
    assert two["energy"]["n_systems"] == 2 * 65536 and two["scaling"] == "weak"
    assert two["gathered_bytes"] == 2 * (65536 * 3 * 8 * 8 + 65536 * 2 * 8 + 65536 * 2 * 4)
    one = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--span", "0.2", "--no-cpu-baseline", "--no-secondary")
    assert one["n_gpus"] == 1 and "gather_ms" not in one and one["energy"]["n_systems"] == 65536
    # two ranks time-share the one GPU: the aggregate cannot be far above one rank's rate, nor below half of it
    assert 0.3 * one["value"] < two["value"] < 2.5 * one["value"]
