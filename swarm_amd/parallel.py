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


# ---------------------------------------------------------------------------------------------------------------------
# The sharded run (BASELINE configs[3]: mvs, nsys = 2^20, nbod = 8 over 8 GPUs).  Every rank owns the same number of
# wave chunks (`capacity`; the last ranks may hold fewer real systems), integrates them with no exchange at all, and
# then takes part in two exchanges: one all_gather_into_tensor of an 11-word vector (the 8 doubles of
# swarm_hip_energy_stats + one checksum per device buffer), and one gather to rank 0 per device buffer (body, sysd,
# sysi) into slices of a preallocated image.  Because a shard is a contiguous run of wave chunks, the gathered buffers
# ARE the wave-chunk image of the whole ensemble: no unpacking step follows.  Only rank 0 receives the image (north_star:
# "the final ensemble gather"): each shard crosses xGMI once, on the link between its GPU and rank 0's -- 7 links into
# rank 0 in parallel; an all-gather would put 8 copies of the 562 MB image of C4 on the fabric for nothing.
#
# The device work goes through a `backend` with the five calls below.  HipShardBackend (the product) makes them C-ABI
# calls on torch-owned device memory; tests/test_parallel_gloo.py supplies a CPU stand-in so that the same harness --
# partition, collectives, verification -- runs under gloo without a GPU.

def buffer_checksums(buffers):
    """one int64 per buffer: the wrapping sum of its bytes read as int64 words (buffer sizes are multiples of 512)"""
    import torch
    return torch.stack([b.view(torch.int64).sum() for b in buffers])


def gather_to_root(buffers, dist, rank, world, dst=0):
    """dist.gather of every buffer into slices of one preallocated tensor per buffer on rank `dst` (RCCL: grouped
    send/recv, the receives land in place).  Returns the list of gathered tensors on `dst`, None elsewhere."""
    import torch
    out = None
    if rank == dst:
        out = [torch.empty(world * b.numel(), dtype=b.dtype, device=b.device) for b in buffers]
    for k, b in enumerate(buffers):
        parts = list(out[k].split(b.numel())) if rank == dst else None
        dist.gather(b, parts, dst=dst)
    return out


class HipShardBackend:
    """One rank's shard on its GPU: torch owns the bytes, libswarm_hip.so does the work."""

    def __init__(self, capacity, nsys, nbod, device, nbattr=1, nsattr=1):
        import torch
        from . import capi
        if not torch.cuda.is_available():
            # the PyTorch wheel and /opt/rocm each ship a libamdhip64.so.7; the first one loaded serves the whole
            # process, and torch finds no GPU behind /opt/rocm's
            raise RuntimeError("torch sees no GPU: in a process that uses both, import torch (and touch torch.cuda) "
                               "BEFORE swarm_amd loads libswarm_hip.so")
        from .ensemble import DeviceEnsemble
        L = capi.lib
        self.capacity, self.nsys, self.nbod, self.nbattr, self.nsattr, self.device = capacity, nsys, nbod, nbattr, nsattr, device
        sizes = (L.swarm_hip_body_bytes(capacity, nbod, nbattr), L.swarm_hip_sysd_bytes(capacity, nsattr),
                 L.swarm_hip_sysi_bytes(capacity))
        self.buffers = [torch.zeros(max(n, 8), dtype=torch.uint8, device=device) for n in sizes]
        # lanes this rank does not own: DISABLED (-1), id -1 -- all-ones bytes in the int32 buffer
        self.buffers[2].fill_(0xFF)
        self.dens = DeviceEnsemble(nsys, nbod, nbattr, nsattr, pointers=[b.data_ptr() for b in self.buffers])
        self._e0 = None

    def upload(self, host):
        if self.nsys:
            self.dens.upload(host)
            self._e0 = self._energy()

    def _energy(self):
        import torch
        from . import capi
        n = capi.lib.swarm_hip_nchunk(self.nsys) * CHUNK
        t = torch.empty(max(n, 1), dtype=torch.float64, device=self.device)
        capi.check(capi.lib.swarm_hip_total_energy(self.dens.ref, t.data_ptr(), None))
        return t

    def integrate(self, cfg, destination_time):
        from .integrator import Integrator
        if not self.nsys:
            return
        integ = Integrator.create(cfg)
        integ.set_ensemble(None, self.dens)
        integ.set_destination_time(destination_time)
        integ.core_integrate()

    def stats(self):
        """device tensor of 8 doubles: {sum, max, min of |dE/E0|, n_active, n_inactive, n_disabled, n_other, nsys}"""
        import torch
        from . import capi
        out = torch.zeros(8, dtype=torch.float64, device=self.device)
        out[2] = float("inf")
        if self.nsys:
            e1 = self._energy()
            capi.check(capi.lib.swarm_hip_energy_stats(self.dens.ref, self._e0.data_ptr(), e1.data_ptr(), out.data_ptr(), None))
            capi.check(capi.lib.swarm_hip_synchronize(None))
        return out

    def download_image(self, buffers, nsys):
        """host ensemble of the first nsys systems of a wave-chunk image held in three device tensors"""
        from .ensemble import DeviceEnsemble
        return DeviceEnsemble(nsys, self.nbod, self.nbattr, self.nsattr, pointers=[b.data_ptr() for b in buffers]).download()


def combine_stats(rows):
    """[world, 8] statistics vectors -> one dict (sum / max / min / counts)"""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 8)
    out = {"sum_abs_dE_over_E": float(rows[:, 0].sum()), "max_abs_dE_over_E": float(rows[:, 1].max()),
           "min_abs_dE_over_E": float(rows[:, 2].min())}
    for k, name in enumerate(("n_active", "n_inactive", "n_disabled", "n_other", "nsys")):
        out[name] = float(rows[:, 3 + k].sum())
    out["mean_abs_dE_over_E"] = out["sum_abs_dE_over_E"] / max(out["nsys"], 1.0)
    return out


def run_sharded(nsys_total, make_shard, make_backend, cfg, destination_time, dist=None, rank=0, world=1,
                sync=lambda: None, clock=None, verify=True):
    """Partition, integrate, exchange, verify.  make_shard(lo, hi) -> host ensemble of systems [lo, hi) with ids
    lo..hi-1; make_backend(capacity, nsys) -> backend.  Returns a dict with the combined statistics, the integrate
    and gather times of this rank, and (rank 0, verify=True) the outcome of the checks on the gathered image:
    `checksums_match` (every rank's slice of the image has the checksum that rank computed of its own buffers),
    `ids_are_arange`, and the image as a host ensemble (`full`)."""
    import time
    import torch
    clock = clock or time.perf_counter
    bounds = shard_bounds(nsys_total, world)
    lo, hi = bounds[rank]
    capacity = max(h - l for l, h in bounds)
    capacity = (capacity + CHUNK - 1) // CHUNK * CHUNK
    be = make_backend(capacity, hi - lo)
    shard = make_shard(lo, hi)
    be.upload(shard)
    multi = dist is not None and dist.is_initialized()
    sync()
    if multi:
        dist.barrier()
    t0 = clock()
    be.integrate(cfg, destination_time)
    sync()
    t_integrate = clock() - t0
    # statistics + buffer checksums: one collective of 11 words (the checksums travel as int64 bit patterns)
    mine = torch.cat([be.stats(), buffer_checksums(be.buffers).to(torch.int64).view(torch.float64)])
    if multi:
        allwords = torch.empty(world * 11, dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(allwords, mine)
    else:
        allwords = mine
    allwords = allwords.reshape(-1, 11)
    stats = combine_stats(allwords[:, :8].cpu().numpy())
    sums = allwords[:, 8:].contiguous().view(torch.int64)       # [world, 3]
    # final shards: one gather per buffer into slices of a preallocated image of world * capacity systems on rank 0
    sync()
    t0 = clock()
    gathered = gather_to_root(be.buffers, dist, rank, world) if multi else be.buffers
    sync()
    t_gather = clock() - t0
    res = {"stats": stats, "integrate_s": t_integrate, "gather_s": t_gather, "bounds": bounds, "capacity": capacity,
           "gathered_bytes": int(sum(b.numel() for b in be.buffers)) * world, "exchange": "gather to rank 0" if multi else "none"}
    if verify and rank == 0:
        # the bytes every rank sent are the bytes at its offset of the image: checksum of each slice against the sender's
        got = torch.stack([buffer_checksums(list(g.split(b.numel()))) for g, b in zip(gathered, be.buffers)], dim=1)
        res["checksums_match"] = bool(torch.equal(got.cpu(), sums.cpu()))
        full = be.download_image(gathered, nsys_total)
        res["ids_are_arange"] = bool((full.id == np.arange(nsys_total)).all())
        res["full"] = full
    return res
