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


def test_rccl_backend_single_rank():
    """The real device transport (torch.distributed "nccl" == RCCL) at the one rank a 1-GPU box allows."""
    env = dict(os.environ)
    env.update(OMP_NUM_THREADS="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29711", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("CRPSPMM_EXCHANGE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_nccl_worker.py")], capture_output=True, text=True,
                       env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GPU_NCCL_WORKER_OK" in r.stdout


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.parametrize("world", [2, 4, 8])
def test_engines_native_rccl_multi_gpu(world):
    """The native RCCL members (grouped ncclSend / ncclRecv exchange, device all-gather of A, the non-blocking bootstrap) with
    one rank per GPU -- the path that has only ever run at world size 1, because every box this suite has seen so far had one
    GPU.  Skipped there; on a node with `world` GPUs it runs the same worker as the host-staged rehearsal, WITHOUT
    CRPSPMM_EXCHANGE=host, and the worker asserts that the device transport is the one in use."""
    if _gpu_count() < world:
        pytest.skip("needs %d GPUs (native RCCL refuses two ranks on one device)" % world)
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["CRPSPMM_EXPECT_NATIVE_RCCL"] = "1"
    env.pop("CRPSPMM_EXCHANGE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29800 + world), os.path.join(ROOT, "tests", "gpu_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GPU_DIST_WORKER_OK world=%d" % world in r.stdout
