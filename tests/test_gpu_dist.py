"""GPU: the N > 1 engines with several ranks sharing the one GPU of the test box (host-staged
exchange; see tests/gpu_dist_worker.py).  At most 4 processes touch the card."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4])
def test_engines_multi_rank_one_gpu(world):
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    env["CRPSPMM_EXCHANGE"] = "host"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), os.path.join(ROOT, "tests", "gpu_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GPU_DIST_WORKER_OK world=%d" % world in r.stdout
