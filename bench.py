#!/usr/bin/env python3
"""Headline benchmark: system-steps/s of the fused Hermite kernel (BASELINE.json configs[1]:
Hermite.cfg fixed step time_step=1e-4, nsys=65536 nbod=3 per GPU).

A bench "step" = one pass of the hot path over the resident ensemble: destination_time advances
by --span (default 2.0 = test/benchmark/Hermite.cfg's destination_time => 20000 Hermite steps per
system), systems are re-activated and ONE kernel launch integrates every system to the new
destination, followed by the reference's active-count check (swarm/integrator.cpp:102-117
without the H2D/D2H copies: inputs are resident in HBM).

N > 1: one process per GPU, the ensemble is partitioned by system index, each rank integrates its own shard (no
data-path collective); after the timed region the energy/ejection statistics are all-reduced and the final shards
gathered to rank 0 over RCCL, outside the timing.  `python bench.py --gpus N` starts the N ranks itself (child
processes through torch.distributed.run, before this process touches torch or the GPU); under an external
torchrun / torch.distributed.run (RANK set) it is one of the ranks.

At N = 1 the JSON line also carries a "lane_per_body" block (three small measurements of the DESIGN 3.9 kernels beside
the lane-per-system ones) and a "secondary" block: BASELINE's other single-GPU configurations (C3
rkck_adaptive 65536 x 6 in its cruising and its divergent regime, one C4 shard of mvs 131072 x 8, C5
hermite_adap_log 65536 x 4), one or two launches each, kernel time by HIP events (tools/perf.py measure()).

--config c4 runs BASELINE configs[3] itself: mvs, time_step = 0.01, nsys = 2^20, nbod = 8 partitioned over the
ranks (swarm_amd.parallel.run_sharded: shard_bounds, one launch loop per shard, all_gather_into_tensor of the
statistics vector + buffer checksums, a gather of the three device buffers into a preallocated image on rank 0,
verification there).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = {3: 831}  # F_hermite(n) = 118*P + 159*n
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz


def profiled_traffic(tag):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/rN/<tag>_pmc.json, newest round first): FETCH_SIZE (KB, doubled on gfx950 per the MI355X guide) +
    WRITE_SIZE (KB).  None when no profile is committed."""
    base = os.path.join(ROOT, "profiles")
    for rnd in sorted((d for d in os.listdir(base) if d.startswith("r")), reverse=True) if os.path.isdir(base) else []:
        fn = os.path.join(base, rnd, tag + "_pmc.json")
        if os.path.exists(fn):
            c = json.load(open(fn))
            c = c.get("counters", c)
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                out = {"bytes": c["FETCH_SIZE"]["per_launch_avg"] * 1024 * 2 + c["WRITE_SIZE"]["per_launch_avg"] * 1024,
                       "source": os.path.relpath(fn, ROOT), "issued_flop": None}
                ops = [c.get("SQ_INSTS_VALU_%s_F64" % k, {}).get("per_launch_avg") for k in ("ADD", "MUL", "FMA", "TRANS")]
                if None not in ops:
                    # FP64 flop ISSUED per launch: wave-instructions x 64 lanes, FMA counted twice (the launches of this
                    # command run with every lane enabled)
                    out["issued_flop"] = 64.0 * (ops[0] + ops[1] + 2.0 * ops[2] + ops[3])
                return out
    return None


def secondary_block():
    """BASELINE's other single-GPU configurations, one short measurement each (kernel time by HIP events)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import perf
    out = {}
    for tag, workload, kw in (("c3_rkck_adaptive_cruising", "c3", {}),
                              ("c3_rkck_adaptive_divergent", "c3_div", {"shuffle": True}),
                              ("c4_mvs_shard", "c4", {}),
                              ("c5_hermite_adap_log", "c5", {})):
        r = perf.measure(workload, reps=5, host_to_host=False, warm=True, **kw)
        keep = {k: r[k] for k in ("integrator", "nsys", "nbod", "T", "ms", "ms_min_max", "reps", "accepted", "attempted",
                                  "attempts", "accepted_steps_per_s", "attempted_steps_per_s", "flop_per_step",
                                  "algorithmic_tflops", "frac_of_fp64_peak", "lane_utilisation_all_launches",
                                  "kernel_ms_all_launches", "kernel_ms_all_launches_min_max", "launches",
                                  "log_bytes", "flush_ms") if k in r}
        keep["ms_is"] = ("first launch of the run, HIP events on the launch stream, of the MEDIAN repetition of 5 (by kernel "
                         "time over all launches of the run; min and max beside it); the same launch on a copy of the "
                         "ensemble is enqueued directly ahead of the timed one so that it does not start at an idle clock")
        keep["reproduce"] = "python tools/perf.py %s %s--warm --no-host-to-host --reps 5" % (workload, "--shuffle " if kw.get("shuffle") else "")
        out[tag] = keep
    # C5's launch loop with its flush (launch, copy of the 396 MB of records into page-locked memory, active count):
    # the reference's protocol, and the two-log manager with the attempt in four launches
    fl = {}
    for name, batches in (("one_log_synchronous_copy_ms", 0), ("two_logs_4_batches_ms", 4)):
        r = perf.measure_flush("c5", batches, reps=5)
        fl[name] = round(r["ms"], 4)
        fl[name.replace("_ms", "_min_max")] = r["ms_min_max"]
        fl["log_bytes"], fl["dropped"] = r["log_bytes"], r["dropped"]
    fl["ms_is"] = "host to host around core_integrate(), ensemble resident, median of 5 after one dropped run"
    fl["reproduce"] = "python tools/perf.py c5 --flush-batches 0|4 --reps 5"
    out["c5_launch_and_flush"] = fl
    return out


def lane_per_body_block():
    """The lane-per-body kernels beside the lane-per-system ones on the same ensembles: lane_mode=1
    forced against lane_mode=0, the library's own choice (one lane per body in all three cases).  Kernel ms."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import perf
    out = {}
    for tag, workload, kw in (("irk2_65536x6_200_steps", "irk2", dict(nsys=65536, nbod=6, T=2.0)),
                              ("mvs_4096x8_1001_steps", "c4", dict(nsys=4096)),
                              ("rkck_adaptive_2048x6_100_attempts", "c3", dict(nsys=2048))):
        ms = {}
        for name, mode in (("lane_per_system_ms", 1), ("library_choice_ms", 0)):
            ms[name] = perf.measure(workload, reps=3, host_to_host=False, warm=True, lane_mode=mode, **kw)["ms"]
        out[tag] = ms
    return out


def hermite_flops(n):
    return 118 * (n * (n - 1) // 2) + 159 * n


def launch_ranks(n):
    """One process per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <same arguments>`
    as a child process of this (GPU-free) one; rendezvous on 127.0.0.1 at a free port.  Rank 0's JSON line goes to
    this process's stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.run(cmd, env=env).returncode
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--nsys", type=int, default=65536, help="systems per GPU")
    ap.add_argument("--nbod", type=int, default=3)
    ap.add_argument("--time-step", type=float, default=1e-4)
    ap.add_argument("--span", type=float, default=2.0, help="destination_time advance per bench step")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=18.0)
    ap.add_argument("--config", choices=["c2", "c4"], default="c2",
                    help="c2: the headline (Hermite.cfg, 65536 x 3 per GPU); c4: mvs 2^20 x 8 sharded over the ranks")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary block (N = 1 only)")
    ap.add_argument("--c4-nsys", type=int, default=1 << 20)
    ap.add_argument("--c4-span", type=float, default=10.0, help="c4: destination_time (1000 steps of 0.01)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 rehearsal on a one-GPU box: every rank uses cuda:0 and the post-run exchange runs "
                         "over gloo on host copies (RCCL refuses two ranks on one device)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks here, as fresh child processes, BEFORE anything
    # in this process imports torch or touches the GPU (a process that has initialised HIP must not be replaced or
    # forked), and exit with their status.  Under torchrun / torch.distributed.run RANK is set and this is a rank.
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)

    import numpy as np
    import torch
    import swarm_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (rank %d of %d; the product path has no CPU fallback)" % (rank, world))
    if args.rehearse_one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants cuda:%d but this node has %d GPU(s) (one process per GPU; "
                         "--rehearse-one-gpu puts every rank on cuda:0)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if args.rehearse_one_gpu else dev   # where collective operands live
    dist, backend = None, None
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running %d rank(s)" % (args.gpus, world, world), file=sys.stderr)
    if world > 1 or "RANK" in os.environ:     # under torchrun the exchange runs even with one rank (RCCL rehearsal)
        import torch.distributed as dist
        backend = "gloo" if args.rehearse_one_gpu else "nccl"
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()
    S.capi.check(S.capi.lib.swarm_hip_init(local_rank))
    L = S.capi.lib
    if args.config == "c4":
        return bench_c4(args, S, torch, dist, dev, cdev, rank, world, backend)

    nsys, nbod = args.nsys, args.nbod
    # global ensemble partitioned by system index: this rank owns [rank*nsys, (rank+1)*nsys)
    cfg = S.config({"nsys": str(nsys), "nbod": str(nbod)})
    host = S.generate_ensemble(cfg)
    host.id[:] = np.arange(rank * nsys, (rank + 1) * nsys)
    # torch owns the device memory (plumbing); the C-ABI gets raw pointers
    tb = torch.empty(L.swarm_hip_body_bytes(nsys, nbod, 1), dtype=torch.uint8, device=dev)
    td = torch.empty(L.swarm_hip_sysd_bytes(nsys, 1), dtype=torch.uint8, device=dev)
    ti = torch.empty(L.swarm_hip_sysi_bytes(nsys), dtype=torch.uint8, device=dev)
    dens = S.DeviceEnsemble(nsys, nbod, pointers=[tb.data_ptr(), td.data_ptr(), ti.data_ptr()])
    dens.upload(host)
    e0 = dens.total_energy()

    integ = S.Integrator.create(S.config({"integrator": "hermite", "time_step": repr(args.time_step)}))
    integ.set_ensemble(host, dens)
    if args.block:
        integ.args.block_size = args.block
    stream = torch.cuda.current_stream().cuda_stream
    integ.stream = stream
    steps_per_pass = int(round(args.span / args.time_step))

    ev = []
    for _ in range(2 * max(args.steps, 1)):
        p = C.c_void_p()
        S.capi.check(L.swarm_hip_event_create(C.byref(p)))
        ev.append(p)

    dest = [0.0]

    def one_pass(k=None):
        dest[0] += args.span
        integ.set_destination_time(dest[0])
        dens.reactivate(stream)
        if k is not None:
            S.capi.check(L.swarm_hip_event_record(ev[2 * k], stream))
        integ.launch_integrator()
        if k is not None:
            S.capi.check(L.swarm_hip_event_record(ev[2 * k + 1], stream))
        n_active = dens.count_active(stream)  # synchronises the stream
        assert n_active == 0, n_active

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_pass(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = []
    for k in range(args.steps):
        ms = C.c_float()
        S.capi.check(L.swarm_hip_event_elapsed_ms(ev[2 * k], ev[2 * k + 1], C.byref(ms)))
        kernel_ms.append(ms.value)
    kernel_ms_avg = sum(kernel_ms) / max(len(kernel_ms), 1)

    # ---- post-run checks and the multi-GPU exchange (outside the timed region) ----
    from swarm_amd import parallel as P
    e1 = dens.total_energy()
    derr = np.abs((e1 - e0) / e0)
    final = dens.download()
    expect_t = (args.warmup + args.steps) * args.span
    assert np.allclose(final.time, expect_t, rtol=1e-9), (final.time[:3], expect_t)
    # energy / ejection statistics: one small all-reduce (sum, max, min) over RCCL
    stats = P.allreduce_stats(*P.local_stats(derr, final.state, steps_accepted=nsys * steps_per_pass * args.steps),
                              dist=dist, device=cdev)
    gather_ms, gather_ok, gather_error = None, None, None
    if dist is not None:
        # final ensemble gather: a shard is a contiguous run of wave chunks, so ONE gather per device buffer into
        # slices of a preallocated tensor on rank 0 gives the wave-chunk image of the whole ensemble; the senders'
        # checksums travel in one small all-gather and rank 0 checks every slice against them.  Outside the timed
        # region: a failure here is reported in the line (gather_verified false + the message), it does not void `value`.
        try:
            srcs = [t.to(cdev) for t in (tb, td, ti)]
            mine = P.buffer_checksums(srcs)
            sums = torch.empty(world * 3, dtype=torch.int64, device=cdev)
            dist.all_gather_into_tensor(sums, mine)
            torch.cuda.synchronize()
            dist.barrier()
            g0 = time.perf_counter()
            outs = P.gather_to_root(srcs, dist, rank, world)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            if rank == 0:
                got = torch.stack([P.buffer_checksums(list(o.split(t.numel()))) for o, t in zip(outs, srcs)], dim=1)
                gather_ok = bool(torch.equal(got.cpu(), sums.reshape(world, 3).cpu()))
                ids = outs[2].view(torch.int32).reshape(-1, 2, 64)[:, 1, :].reshape(-1).cpu().numpy()
                gather_ok = gather_ok and bool((ids == np.arange(world * nsys)).all())
        except Exception as ex:     # noqa: BLE001 -- reported, not swallowed
            gather_ok, gather_error = False, "%s: %s" % (type(ex).__name__, ex)

    total_sys_steps = float(world) * nsys * steps_per_pass * args.steps
    value = total_sys_steps / elapsed
    flops = hermite_flops(nbod)
    achieved = flops * nsys * steps_per_pass / (kernel_ms_avg * 1e-3) / 1e12 if kernel_ms_avg > 0 else 0.0

    traffic = profiled_traffic("hermite_n3") if (nbod, nsys) == (3, 65536) else None
    out = {
        "metric": "system-steps/sec, Hermite nbod=3 nsys=64k" if (nbod, nsys) == (3, 65536)
        else "system-steps/sec, Hermite nbod=%d nsys=%d" % (nbod, nsys),
        "value": value,
        "unit": "system-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "Hermite.cfg fixed-step time_step=%g, nsys=%d nbod=%d per GPU, "
                               "generate_ensemble systems, %d Hermite steps per system per bench step"
                               % (args.time_step, nsys, nbod, steps_per_pass),
                   "integrator": "hermite", "nsys_per_gpu": nsys, "nbod": nbod, "time_step": args.time_step,
                   "partition": "by system index, %d shard(s), no data-path collective" % world},
        "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                     # HBM bytes per launch from the PMC passes of this same command (FETCH_SIZE doubled per
                     # the MI355X guide + WRITE_SIZE), profiles/r1/hermite_n3_pmc.json; algorithmic = 340 B/system
                     "traffic": traffic["bytes"] if traffic else None,
                     "traffic_source": traffic["source"] if traffic else None,
                     # FP64 flop the kernel ISSUES (SQ_INSTS_VALU_{ADD,MUL,FMA x 2,TRANS}_F64 x 64 lanes per launch, from
                     # the same committed PMC passes) over this run's kernel time, as a fraction of the same peak:
                     # `frac` counts the reference's formulas, this counts the instructions
                     "issued_frac": (traffic["issued_flop"] / (kernel_ms_avg * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS
                                     if traffic and traffic.get("issued_flop") and kernel_ms_avg > 0 else None),
                     "kernel": "hermite_kernel<%d,false,0>" % nbod,
                     "kernel_ms_avg": kernel_ms_avg,
                     "flop_per_system_step": flops,
                     "note": "bound is the FP64 vector ALU (not HBM, not MFMA): state is register-resident for "
                             "all %d steps of a launch; HBM bytes per launch are negligible" % steps_per_pass},
        "energy": {"max_rel_err": stats["max_abs_dE_over_E"], "mean_rel_err": stats["mean_abs_dE_over_E"],
                   "n_disabled": int(stats["n_disabled"]), "n_systems": int(stats["nsys"])},
    }
    if dist is not None:
        out["gather_ms"] = gather_ms
        out["gather_verified"] = bool(gather_ok)
        if gather_error:
            out["gather_error"] = gather_error
        out["gathered_bytes"] = int(world * (tb.numel() + td.numel() + ti.numel()))
        out["exchange"] = "%s: all_reduce x 3 of the statistics, all_gather of 3 checksums, gather of body/sysd/sysi to rank 0" % backend
        out["rccl_ranks"] = world if backend == "nccl" else 0
    if rank == 0 and world == 1 and dist is None and not args.no_secondary:
        out["secondary"] = secondary_block()
        out["lane_per_body"] = lane_per_body_block()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        threads = min(16, O.max_threads())
        cn = 1024 * threads
        ce = O.generate(cn, nbod)
        # calibrate then run ~cpu_seconds of work
        t0 = time.perf_counter()
        O.integrate(ce, O.params("hermite_cpu", 0.02, time_step=args.time_step), threads=threads)
        probe = time.perf_counter() - t0
        rate = cn * 200 / probe
        T = max(0.05, min(20.0, args.cpu_seconds * rate / cn * args.time_step))
        T = round(T / args.time_step) * args.time_step
        ce = O.generate(cn, nbod)
        t0 = time.perf_counter()
        acc, _ = O.integrate(ce, O.params("hermite_cpu", T, time_step=args.time_step), threads=threads, counters=True)
        ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(acc.sum()) / ct, "unit": "system-steps/s", "cores": threads,
                               "kind": "port",
                               "sample": "oracle hermite_cpu (C restatement of integrators/hermite_cpu.hpp, OpenMP), "
                                         "nsys=%d nbod=%d time_step=%g, %d steps/system, %.1f s"
                                         % (cn, nbod, args.time_step, int(acc[0]), ct)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_c4(args, S, torch, dist, dev, cdev, rank, world, backend):
    """BASELINE configs[3]: mvs, nsys = 2^20, nbod = 8, time_step = 0.01, partitioned over the ranks.  One bench step
    = the whole run (destination_time = --c4-span); `value` = system-steps of all ranks / the slowest rank's time."""
    import numpy as np
    from swarm_amd import parallel as P
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import perf
    nsys_total, nbod = args.c4_nsys, 8
    cfg = S.config({"integrator": "mvs", "time_step": "0.01"})

    def make_shard(lo, hi):
        h = S.generate_ensemble(S.config({"nsys": str(max(hi - lo, 1)), "nbod": str(nbod)}))
        if hi - lo == 0:
            h = h.slice(0, 0)
        h.id[:] = np.arange(lo, hi)
        return h
    if args.rehearse_one_gpu:
        raise SystemExit("--config c4 exchanges device buffers over RCCL: one rank per GPU")
    times = []
    res = None
    for k in range(args.warmup + args.steps):
        res = P.run_sharded(nsys_total, make_shard, lambda cap, n: P.HipShardBackend(cap, n, nbod, dev), cfg,
                            args.c4_span, dist=dist, rank=rank, world=world, sync=torch.cuda.synchronize,
                            verify=(k == args.warmup + args.steps - 1))
        t = torch.tensor([res["integrate_s"], res["gather_s"]], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if k >= args.warmup:
            times.append(t.cpu().numpy())
    integ_s = float(np.mean([t[0] for t in times]))
    gather_s = float(np.mean([t[1] for t in times]))
    steps_per_system = int(round(args.c4_span / 0.01)) + 1      # the 1001st step closes the rounding gap to T
    total = float(nsys_total) * steps_per_system
    flops = perf.mvs_flops(nbod)
    achieved = flops * total / world / integ_s / 1e12
    assert rank != 0 or (res["checksums_match"] and res["ids_are_arange"]), "gathered image differs from the shards"
    if rank == 0:
        full = res["full"]
        assert (full.state == 1).all() and np.allclose(full.time, args.c4_span, rtol=1e-12)
        print(json.dumps({
            "metric": "system-steps/sec, MVS nbod=8 nsys=%d over %d GPU(s)" % (nsys_total, world),
            "value": total / integ_s, "unit": "system-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": integ_s * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "MVS.cfg time_step=0.01, nsys=%d nbod=8, generate_ensemble systems, destination_time=%g "
                                   "(%d steps per system), shards of %d systems" % (nsys_total, args.c4_span, steps_per_system,
                                                                                    res["capacity"]),
                       "integrator": "mvs", "partition": "by system index, %d shard(s), no data-path collective" % world},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": None,
                         "kernel": "generic_kernel<8,MVS,0>", "flop_per_system_step": flops,
                         "note": "per GPU; the time is host to host over the shard's launch loop"},
            "exchange": {"gather_ms": gather_s * 1e3, "gathered_bytes": res["gathered_bytes"],
                         "collectives": "all_gather_into_tensor of 11 words (statistics + 3 checksums), gather to rank 0 "
                                        "x 3 (body, sysd, sysi)" if dist is not None else "none (one rank)",
                         "rccl_ranks": world if backend == "nccl" else 0,
                         "checksums_match": True, "ids_are_arange": True},
            "energy": {"max_rel_err": res["stats"]["max_abs_dE_over_E"], "mean_rel_err": res["stats"]["mean_abs_dE_over_E"],
                       "n_inactive": int(res["stats"]["n_inactive"]), "n_systems": int(res["stats"]["nsys"])},
        }), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
