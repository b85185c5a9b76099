import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built artefacts are git-ignored.  In a tree where they are missing (a fresh checkout) and a
    # compiler exists, build them once (hipcc cross-compiles gfx950 without a GPU); on the GPU box the
    # prebuilt files travel with the tree and nothing is built.
    need = [os.path.join(ROOT, "swarm_amd", "lib", "libswarm_hip.so"), os.path.join(ROOT, "swarm_amd", "lib", "libswarmng_hip.so"),
            os.path.join(ROOT, "swarm_amd", "bin", "swarm"), os.path.join(ROOT, "swarm_amd", "bin", "swarm_nattr3"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(f) for f in need) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu():
    """swarm_amd with device 0 selected.  Fails loudly when the HIP extension is missing."""
    import swarm_amd as S
    S.capi.check(S.capi.lib.swarm_hip_init(0))
    return S


def to_host(S, e):
    """oracle.Ensemble -> swarm_amd.HostEnsemble (deep copy)"""
    h = S.HostEnsemble(e.nsys, e.nbod, e.body_attr.shape[2], e.sys_attr.shape[1])
    for k in ("pos", "vel", "mass", "time", "state", "id", "body_attr", "sys_attr"):
        setattr(h, k, getattr(e, k).copy())
    return h


def rel_diff(a, b):
    """max over bodies of |a-b| / max(|b|, tiny): per-body Euclidean, relative to the body's norm"""
    d = np.sqrt(((a - b) ** 2).sum(-1))
    n = np.sqrt((b ** 2).sum(-1))
    return float((d / np.maximum(n, 1e-300)).max())


def perturbed(O, nsys, nbod, seed=1, amp=1e-3, **kw):
    """Generated ensemble with a deterministic per-system perturbation so that systems differ."""
    e = O.generate(nsys, nbod, **kw)
    rng = np.random.default_rng(seed)
    e.pos[:, 1:, :] *= 1.0 + amp * rng.standard_normal((nsys, nbod - 1, 3))
    e.vel[:, 1:, :] *= 1.0 + amp * rng.standard_normal((nsys, nbod - 1, 3))
    e.pos[:, 1:, 2] += amp * rng.standard_normal((nsys, nbod - 1))
    return e


def scaled(O, nsys, nbod, lo, hi, seed=3, amp=1e-2, **kw):
    """Perturbed ensemble whose systems are geometrically scaled copies (size factor lo..hi,
    velocities / sqrt(factor)): dynamical times differ by (hi/lo)^1.5 across the ensemble, which
    makes adaptive integrators take very different step counts per system."""
    e = perturbed(O, nsys, nbod, seed=seed, amp=amp, **kw)
    s = np.geomspace(lo, hi, nsys)[:, None, None]
    e.pos *= s
    e.vel /= np.sqrt(s)
    return e


def head_on(O, nsys, planet_mass=1e-3):
    """Two planets on (almost) the same circle moving towards each other; the initial angular
    separation varies per system, so the close-encounter time does too."""
    e = O.generate(nsys, 3, planet_mass=planet_mass)
    r = 10.0
    v = np.sqrt(1 / r)
    dth = np.linspace(0.2, 1.5, nsys)
    z = 0 * dth
    e.pos[:, 1] = np.stack([r + z, z, z], -1)
    e.vel[:, 1] = np.stack([z, v + z, z], -1)
    e.pos[:, 2] = np.stack([1.02 * r * np.cos(dth), 1.02 * r * np.sin(dth), z], -1)
    e.vel[:, 2] = np.stack([v * np.sin(dth), -v * np.cos(dth), z], -1)
    return e
