"""bench.py's own rank launcher, without a GPU: `python bench.py --gpus 2` must start two ranks as child processes
(torch.distributed.run) and hand their status back.  Here every rank stops at "needs a GPU", which is the evidence:
the message names the rank and the world size of 2, and the exit status is the launcher's failure status."""
import os
import subprocess
import sys

from conftest import ROOT


def test_bench_gpus_n_starts_n_ranks_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check of the launcher; tests/test_gpu_sharded.py covers the GPU box")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # (the elastic agent ends the other rank as soon as the first has failed: that one may not get to its own message)
    said = p.stderr.count("bench.py needs a GPU")
    assert 1 <= said <= 2 and " of 2;" in p.stderr, p.stderr[-3000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]      # no JSON line from a failed run
