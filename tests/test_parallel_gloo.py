"""The N>1 path on CPU: world_size-2 (and 3) gloo processes exercise the partition, the
statistics all-reduce and the final gather.  No kernels run here (no GPU); each rank "integrates"
its shard with a stand-in transformation so that the gathered result is checkable."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nsys, nbod, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import swarm_amd as S
    from swarm_amd import parallel as P
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
    rng = np.random.default_rng(7)
    full.pos += rng.standard_normal(full.pos.shape)
    full.state[:] = rng.integers(-1, 3, nsys)
    shard = P.take_shard(full, rank, world)
    lo, hi = P.shard_bounds(nsys, world)[rank]
    assert (shard.id == np.arange(lo, hi)).all()
    # stand-in for the integration of this shard
    shard.pos *= 2.0
    shard.time += 1.0 + shard.id
    derr = np.abs(np.sin(shard.id + 1.0)) * 1e-12
    stats = P.allreduce_stats(*P.local_stats(derr, shard.state, n_log_records=3 * shard.nsys,
                                             steps_accepted=10 * shard.nsys), dist=dist)
    got = P.gather_ensemble(shard, nsys, dist=dist)
    ok = (np.array_equal(got.pos, full.pos * 2.0) and np.array_equal(got.vel, full.vel)
          and np.array_equal(got.time, 1.0 + np.arange(nsys)) and np.array_equal(got.state, full.state)
          and np.array_equal(got.id, np.arange(nsys)) and np.array_equal(got.mass, full.mass))
    all_err = np.abs(np.sin(np.arange(nsys) + 1.0)) * 1e-12
    ok = ok and abs(stats["sum_abs_dE_over_E"] - all_err.sum()) < 1e-22 and stats["max_abs_dE_over_E"] == all_err.max()
    ok = ok and stats["min_abs_dE_over_E"] == all_err.min() and stats["nsys"] == nsys
    ok = ok and stats["n_disabled"] == (full.state == -1).sum() and stats["n_active"] == (full.state == 0).sum()
    ok = ok and stats["n_log_records"] == 3 * nsys and stats["steps_accepted"] == 10 * nsys
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nsys", [(2, 1000), (2, 64), (3, 777)])
def test_partition_allreduce_gather(world, nsys):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nsys, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_shard_bounds_properties():
    from swarm_amd import parallel as P
    for nsys in (1, 63, 64, 65, 1000, 65536, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            b = P.shard_bounds(nsys, world)
            assert b[0][0] == 0 and b[-1][1] == nsys
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(lo % 64 == 0 or lo == hi == nsys for lo, hi in b)
    assert P.shard_bounds(1 << 20, 8) == [(r * 131072, (r + 1) * 131072) for r in range(8)]


# ---- the sharded C4 harness (swarm_amd.parallel.run_sharded) under gloo, with a CPU stand-in for the device ---------
def _pack(h, capacity):
    """numpy restatement of the wave-chunk image (include/swarm_hip.h): body [nchunk][nbod][7+nbattr][64],
    sysd [nchunk][1+nsattr][64], sysi [nchunk][2][64]; lanes beyond nsys: zeros, state -1, id -1"""
    nch, nb, nba, nsa = capacity // 64, h.nbod, h.nbattr, h.nsattr
    body = np.zeros((nch, nb, 7 + nba, 64))
    sysd = np.zeros((nch, 1 + nsa, 64))
    sysi = np.full((nch, 2, 64), -1, dtype=np.int32)
    n = h.nsys
    k, l = np.arange(n) // 64, np.arange(n) % 64
    for b in range(nb):
        for c in range(3):
            body[k, b, c, l] = h.pos[:, b, c]
            body[k, b, 3 + c, l] = h.vel[:, b, c]
        body[k, b, 6, l] = h.mass[:, b]
        for a in range(nba):
            body[k, b, 7 + a, l] = h.body_attr[:, b, a]
    sysd[k, 0, l] = h.time
    for a in range(nsa):
        sysd[k, 1 + a, l] = h.sys_attr[:, a]
    sysi[k, 0, l], sysi[k, 1, l] = h.state, h.id
    return body, sysd, sysi


def _unpack(S, body, sysd, sysi, nsys, nbod, nba=1, nsa=1):
    body = np.asarray(body).view(np.float64).reshape(-1, nbod, 7 + nba, 64)
    sysd = np.asarray(sysd).view(np.float64).reshape(-1, 1 + nsa, 64)
    sysi = np.asarray(sysi).view(np.int32).reshape(-1, 2, 64)
    h = S.HostEnsemble(nsys, nbod, nba, nsa)
    k, l = np.arange(nsys) // 64, np.arange(nsys) % 64
    for b in range(nbod):
        for c in range(3):
            h.pos[:, b, c] = body[k, b, c, l]
            h.vel[:, b, c] = body[k, b, 3 + c, l]
        h.mass[:, b] = body[k, b, 6, l]
        for a in range(nba):
            h.body_attr[:, b, a] = body[k, b, 7 + a, l]
    h.time[:] = sysd[k, 0, l]
    for a in range(nsa):
        h.sys_attr[:, a] = sysd[k, 1 + a, l]
    h.state[:], h.id[:] = sysi[k, 0, l], sysi[k, 1, l]
    return h


class _CpuStandIn:
    """run_sharded's backend without a GPU: the three buffers are CPU byte tensors holding the same wave-chunk
    image, and the shard is integrated by the oracle's MVS (test infrastructure standing in for the kernel)."""

    def __init__(self, capacity, nsys, nbod):
        import torch
        import swarm_amd as S
        self.S, self.capacity, self.nsys, self.nbod = S, capacity, nsys, nbod
        empty = S.HostEnsemble(0, nbod)
        self.buffers = [torch.from_numpy(a.reshape(-1).view(np.uint8).copy()) for a in _pack(empty, capacity)]

    def _store(self, h):
        import torch
        for t, a in zip(self.buffers, _pack(h, self.capacity)):
            t.copy_(torch.from_numpy(a.reshape(-1).view(np.uint8)))

    def _load(self):
        return _unpack(self.S, *[b.numpy() for b in self.buffers], self.nsys, self.nbod)

    def upload(self, host):
        self._store(host)
        self.e0 = self._load().calc_total_energy()

    def integrate(self, cfg, destination_time):
        from oracle import oracle as O
        h = self._load()
        e = O.generate(max(h.nsys, 1), h.nbod)
        if h.nsys:
            for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
                setattr(e, k, getattr(h, k).copy())
            O.integrate(e, O.params(cfg["integrator"], destination_time, time_step=float(cfg["time_step"])))
            for k in ("pos", "vel", "time", "state"):
                setattr(h, k, getattr(e, k))
        self._store(h)

    def stats(self):
        import torch
        h = self._load()
        if not h.nsys:
            return torch.tensor([0, 0, np.inf, 0, 0, 0, 0, 0], dtype=torch.float64)
        err = np.abs((h.calc_total_energy() - self.e0) / self.e0)
        st = h.state
        return torch.tensor([err.sum(), err.max(), err.min(), (st == 0).sum(), (st == 1).sum(), (st == -1).sum(),
                             ((st != 0) & (st != 1) & (st != -1)).sum(), h.nsys], dtype=torch.float64)

    def download_image(self, buffers, nsys):
        return _unpack(self.S, *[b.numpy() for b in buffers], nsys, self.nbod)


def _sharded_worker(rank, world, port, nsys, nbod, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import swarm_amd as S
    from swarm_amd import parallel as P
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def whole():
        full = S.generate_ensemble(S.config({"nsys": str(nsys), "nbod": str(nbod)}))
        rng = np.random.default_rng(11)
        full.pos[:, 1:] *= 1 + 1e-3 * rng.standard_normal((nsys, nbod - 1, 3))
        return full
    full0 = whole()
    cfg = S.config({"integrator": "mvs", "time_step": "0.01"})
    res = P.run_sharded(nsys, lambda lo, hi: full0.slice(lo, hi), lambda cap, n: _CpuStandIn(cap, n, nbod), cfg, 0.2,
                        dist=dist, rank=rank, world=world)
    ok = res["stats"]["nsys"] == nsys and res["stats"]["n_inactive"] == nsys and res["exchange"] == "gather to rank 0"
    ok = ok and (("full" in res) == (rank == 0))          # only rank 0 receives the image
    ok = ok and res["stats"]["max_abs_dE_over_E"] < 1e-6
    if rank == 0:
        # the gathered image is the unsharded run
        e = O.generate(nsys, nbod)
        for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
            setattr(e, k, getattr(full0, k).copy())
        O.integrate(e, O.params("mvs", 0.2, time_step=0.01))
        got = res["full"]
        ok = ok and res["checksums_match"] and res["ids_are_arange"] and np.array_equal(got.pos, e.pos) and np.array_equal(got.vel, e.vel)
        ok = ok and np.array_equal(got.time, e.time) and np.array_equal(got.state, e.state)
        ok = ok and np.array_equal(got.mass, e.mass)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nsys", [(2, 200), (2, 64), (3, 330)])
def test_sharded_run_harness_under_gloo(world, nsys, oracle):
    """bench.py --config c4 drives swarm_amd.parallel.run_sharded with the HIP backend; here the same function runs
    with a CPU stand-in at world 2 and 3 (ragged shards, an empty rank at nsys=64): every rank's slice of the image gathered
    on rank 0 has that rank's checksum, gathered ids = arange(nsys), gathered ensemble = the unsharded integration, statistics combined."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, nsys, 4, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(r, True) for r in range(world)]
