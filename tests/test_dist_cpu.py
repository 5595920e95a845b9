"""CPU: the N > 1 path over torch.distributed (gloo), world_size 2, 4 and 8 (every grid is forced in turn: world 8
includes the 2 x 4 grid of BASELINE configs[2])."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("world", [2, 4, 8])
def test_engines_over_gloo(world):
    env = dict(os.environ)
    env.pop("RP_SPMM_REIDX", None)
    env["OMP_NUM_THREADS"] = "1"
    port = 29600 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_WORKER_OK world=%d" % world in r.stdout
