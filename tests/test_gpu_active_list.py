"""swarm_hip_launch_t::active_hint: when the caller's attempt loop knows how many systems are still ACTIVE, the choice
between the two kernel layouts goes by that number and a lane-per-body launch works on the ACTIVE systems gathered
into dense wavefronts (active_list_kernel + body_slot_system).  Results must not depend on any of it."""
import numpy as np
import pytest

from conftest import perturbed, scaled, to_host
from test_gpu_integrators import CASES as _CASES

pytestmark = pytest.mark.gpu
CASES = dict(_CASES, hermite=({"time_step": "0.01"}, dict(time_step=0.01), 1.0))


def launch_once(S, h, name, cfg, T, hint, lane_mode=0, log=None):
    """upload, ONE launch with the given active_hint, download; returns (host ensemble, accepted steps)"""
    c = dict(cfg, integrator=name, lane_mode=str(lane_mode))
    integ = S.Integrator.create(S.config(c))
    integ.set_ensemble(h)
    integ.set_destination_time(T)
    integ.enable_step_counters()
    if log is not None:
        integ.set_log_manager(log)
    integ.upload_ensemble()
    integ._dens.last_active_count = hint
    integ.launch_integrator()
    S.capi.check(S.capi.lib.swarm_hip_synchronize(None))
    integ.flush_log()
    integ.download_ensemble()
    return integ.step_counts()[0]


@pytest.mark.parametrize("name", ["hermite", "hermite_adap", "rkck_adaptive", "rkck_fixed", "mvs", "euler", "midpoint",
                                  "verlet", "hermite_prop", "irk2"])
@pytest.mark.parametrize("nbod", [3, 6])
def test_gathered_launch_is_bit_identical(gpu, oracle, name, nbod):
    """1000 systems, 137 of them ACTIVE in no particular pattern (the others DISABLED or INACTIVE): the launch over the
    gathered list gives the ACTIVE systems bit for bit what the ungathered lane-per-body launch and the lane-per-system
    launch give them, leaves the others untouched and counts the same steps."""
    S, O = gpu, oracle
    cfg, _, _ = CASES[name]
    e = perturbed(O, 1000, nbod, seed=40 + nbod)
    rng = np.random.default_rng(9)
    e.state[:] = O.DISABLED
    e.state[1::2] = O.INACTIVE
    on = np.sort(rng.choice(1000, 137, replace=False))
    e.state[on] = O.ACTIVE
    e.time[on[::4]] = 0.03
    T = 0.3
    res = {}
    for tag, hint, lm in (("gathered", 137, 2), ("plain", None, 2), ("auto", 137, 0), ("system", None, 1)):
        h = to_host(S, e)
        acc = launch_once(S, h, name, cfg, T, hint, lm)
        res[tag] = (h, acc)
    base, bacc = res["system"]
    assert (base.state[on] == O.INACTIVE).all() and (bacc[on] > 0).all()
    off = np.setdiff1d(np.arange(1000), on)
    for tag in ("gathered", "plain", "auto"):
        h, acc = res[tag]
        for k in ("pos", "vel", "time", "state"):
            assert np.array_equal(getattr(h, k), getattr(base, k)), (tag, k)
        assert np.array_equal(acc, bacc) and (acc[off] == 0).all()
        assert np.array_equal(h.pos[off], e.pos[off]) and np.array_equal(h.state[off], e.state[off])


def test_hint_below_and_above_the_true_count(gpu, oracle):
    """A hint is a hint: above the true count the surplus workgroups find nothing to do; below it the systems beyond
    the list stay ACTIVE and untouched for the next launch."""
    S, O = gpu, oracle
    cfg, _, _ = CASES["hermite"]
    e = perturbed(O, 640, 4, seed=8)
    e.state[:] = O.DISABLED
    e.state[::5] = O.ACTIVE                 # 128 ACTIVE
    ref = to_host(S, e)
    racc = launch_once(S, ref, "hermite", cfg, 0.2, None, 2)
    h = to_host(S, e)
    acc = launch_once(S, h, "hermite", cfg, 0.2, 500, 2)
    assert np.array_equal(h.pos, ref.pos) and np.array_equal(h.state, ref.state) and np.array_equal(acc, racc)
    assert (racc[::5] >= 20).all() and racc.sum() == racc[::5].sum()
    h = to_host(S, e)
    acc = launch_once(S, h, "hermite", cfg, 0.2, 50, 2)
    done = acc > 0
    assert done.sum() == 50 and (h.state[done] == O.INACTIVE).all()
    rest = (e.state == O.ACTIVE) & ~done
    assert rest.sum() == 78 and (h.state[rest] == O.ACTIVE).all() and np.array_equal(h.pos[rest], e.pos[rest])
    assert np.array_equal(h.pos[done], ref.pos[done])
    acc2 = launch_once(S, h, "hermite", cfg, 0.2, None, 2)      # the next launch finishes them
    assert np.array_equal(h.pos, ref.pos) and np.array_equal(h.state, ref.state)


def test_attempt_loop_hands_the_count_on_and_logs_the_same(gpu, oracle):
    """rkck_adaptive with divergent step counts and ejections over several launches (rkck caps a launch at
    destination_time / max_time_step attempts): integrate() with the hint (default) and without (active_hint=0 in the
    configuration) give the same ensemble bit for bit, the same step counts and the same snapshot records."""
    S, O = gpu, oracle
    e = scaled(O, 3000, 5, 0.03, 1.0, seed=17)
    perm = np.random.default_rng(2).permutation(3000)
    for k in ("pos", "vel", "mass"):
        setattr(e, k, np.ascontiguousarray(getattr(e, k)[perm]))
    cfg = {"integrator": "rkck_adaptive", "min_time_step": "1e-5", "max_time_step": "0.02", "error_tolerance": "1e-24",
           "deactivate_on_ejection": "1", "log_on_ejection": "1", "rmax": "12"}
    out = {}
    for tag, extra in (("hint", {}), ("nohint", {"active_hint": "0"})):
        h = to_host(S, e)
        integ = S.Integrator.create(S.config(dict(cfg, **extra)))
        lm = S.LogManager(8 << 20, "host")
        integ.set_log_manager(lm)
        integ.set_ensemble(h)
        integ.set_destination_time(2.0)
        integ.enable_step_counters()
        n = integ.integrate()
        recs = sorted((r["sys"], r["time"], r["bodies"].tobytes()) for r in O.parse_log(bytes(lm.host)))
        out[tag] = (h, integ.step_counts(), recs, n)
    (h1, c1, r1, n1), (h0, c0, r0, n0) = out["hint"], out["nohint"]
    assert n1 == n0 and n1 > 3
    for k in ("pos", "vel", "time", "state"):
        assert np.array_equal(getattr(h1, k), getattr(h0, k)), k
    assert np.array_equal(c1[0], c0[0]) and np.array_equal(c1[1], c0[1])
    # (a disabled system still beyond rmax logs again at every launch: rkck calls its monitor before the loop for
    # every system whatever its state, rkck_integrator.hpp:140 -- the gathered launches keep that)
    assert r1 == r0 and len(set(r[0] for r in r1)) == (h1.state == O.DISABLED).sum() > 10 and len(r1) > 2 * 617
