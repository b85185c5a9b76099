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
