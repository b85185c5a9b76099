"""Multi-GPU layer: the ensemble is partitioned by system index across ranks (one process per
GPU); systems are independent, so there is NO data-path collective.  The only exchanges are, after
the integration, (1) an all-reduce of a small statistics vector and (2) a gather of the final
shards -- over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests).

The reference has no multi-GPU code at all (only cudaSetDevice); this is the
MI355X-side addition named in BASELINE.json's north_star."""
import numpy as np

from .ensemble import HostEnsemble

CHUNK = 64


def shard_bounds(nsys, world):
    """Contiguous block partition by system index, rounded to whole wave chunks (64 systems) so
    every shard is a contiguous byte range of each device buffer: rank r owns [lo[r], hi[r])."""
    nchunk = (nsys + CHUNK - 1) // CHUNK
    per = (nchunk + world - 1) // world
    bounds = []
    for r in range(world):
        lo = min(nsys, r * per * CHUNK)
        hi = min(nsys, (r + 1) * per * CHUNK)
        bounds.append((lo, hi))
    return bounds


def take_shard(host, rank, world):
    lo, hi = shard_bounds(host.nsys, world)[rank]
    return host.slice(lo, hi)


# statistics vector: sums first, then maxima, then minima
_SUM_FIELDS = ("sum_abs_dE_over_E", "n_active", "n_inactive", "n_disabled", "n_other", "nsys", "n_log_records",
               "steps_accepted", "steps_attempted")


def local_stats(dE_rel, state, n_log_records=0, steps_accepted=0, steps_attempted=0):
    """per-rank statistics from |dE/E0| per system and the state flags"""
    dE_rel = np.asarray(dE_rel, dtype=np.float64)
    state = np.asarray(state)
    sums = np.array([dE_rel.sum(), (state == 0).sum(), (state == 1).sum(), (state == -1).sum(),
                     ((state != 0) & (state != 1) & (state != -1)).sum(), state.size, n_log_records,
                     steps_accepted, steps_attempted], dtype=np.float64)
    mx = np.array([dE_rel.max() if dE_rel.size else 0.0])
    mn = np.array([dE_rel.min() if dE_rel.size else np.inf])
    return sums, mx, mn


def allreduce_stats(sums, mx, mn, dist=None, device=None):
    """energy_conservation_error_range (swarm/utils.cpp:108-121) across ranks: sum / max / min"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        s, a, b = sums, mx, mn
    else:
        import torch
        s = torch.as_tensor(sums, dtype=torch.float64, device=device)
        a = torch.as_tensor(mx, dtype=torch.float64, device=device)
        b = torch.as_tensor(mn, dtype=torch.float64, device=device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(a, op=dist.ReduceOp.MAX)
        dist.all_reduce(b, op=dist.ReduceOp.MIN)
        s, a, b = s.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()
    out = dict(zip(_SUM_FIELDS, (float(x) for x in s)))
    out["max_abs_dE_over_E"] = float(a[0])
    out["min_abs_dE_over_E"] = float(b[0])
    out["mean_abs_dE_over_E"] = out["sum_abs_dE_over_E"] / max(out["nsys"], 1.0)
    return out


def gather_ensemble(shard, nsys_total, dist=None, device=None):
    """Gather the final shards into one HostEnsemble on every rank (all_gather of padded AoS
    blocks; shards differ in size by at most one partition unit)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return shard.clone()
    import torch
    world = dist.get_world_size()
    bounds = shard_bounds(nsys_total, world)
    cap = max(hi - lo for lo, hi in bounds)
    nb = shard.nbod
    width = nb * 3 * 2 + nb + 1 + nb * shard.nbattr + shard.nsattr + 2   # doubles per system (+state,id)
    blk = np.zeros((cap, width))
    n = shard.nsys
    cols = [shard.pos.reshape(n, nb * 3), shard.vel.reshape(n, nb * 3), shard.mass, shard.time[:, None],
            shard.body_attr.reshape(n, nb * shard.nbattr), shard.sys_attr, shard.state[:, None].astype(np.float64),
            shard.id[:, None].astype(np.float64)]
    blk[:n] = np.concatenate(cols, axis=1)
    t = torch.as_tensor(blk, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    full = HostEnsemble(nsys_total, nb, shard.nbattr, shard.nsattr)
    for r, (lo, hi) in enumerate(bounds):
        a = outs[r].cpu().numpy()[: hi - lo]
        k = 0

        def cut(w):
            nonlocal k
            v = a[:, k:k + w]
            k += w
            return v
        full.pos[lo:hi] = cut(nb * 3).reshape(-1, nb, 3)
        full.vel[lo:hi] = cut(nb * 3).reshape(-1, nb, 3)
        full.mass[lo:hi] = cut(nb)
        full.time[lo:hi] = cut(1)[:, 0]
        full.body_attr[lo:hi] = cut(nb * shard.nbattr).reshape(-1, nb, shard.nbattr)
        full.sys_attr[lo:hi] = cut(shard.nsattr)
        full.state[lo:hi] = cut(1)[:, 0].astype(np.int32)
        full.id[lo:hi] = cut(1)[:, 0].astype(np.int32)
    return full
