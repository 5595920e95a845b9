"""GPU: the MPI-typed facade (lib/libcrpspmm.so: rp_spmm_*, para2d_spmm_*, mat_redist_engine_*)
driven end to end by the example programs, which keep the reference's command lines and output
lines (examples/test_rp_spmm.c, test_para2d_spmm.c).  Ranks share the one GPU of the test box;
the facade stages the B exchange through the host (plain MPI), every kernel is the real one."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"


def _run(exe, np_, mtx, n, extra_env=None, tail=("2", "0", "1")):
    path = os.path.join(ROOT, "examples", exe)
    if not os.path.exists(path) or not os.path.exists(MPIEXEC):
        pytest.skip("no MPI launcher / example drivers not built on this machine")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env["PATH"] = os.path.dirname(MPIEXEC) + ":" + env["PATH"]
    env.update(extra_env or {})
    r = subprocess.run([MPIEXEC, "-np", str(np_), path, os.path.join(GOLDEN, mtx), str(n), *tail],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    m = re.search(r"\|\|C_ref - C\|\|_f / \|\|C_ref\|\|_f = ([0-9.eE+-]+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert float(m.group(1)) <= 1e-12
    return r.stdout


@pytest.mark.parametrize("np_", [1, 2, 3])
def test_rp_spmm_driver(np_):
    out = _run("test_rp_spmm.exe", np_, "g_symm.mtx", 33)
    assert "Total rp_spmm_exec()" in out and "Using naive 1D row partitioning" in out
    out = _run("test_rp_spmm.exe", np_, "g_gen.mtx", 8)          # non-square: B rows split evenly
    assert "A size = 120 * 200" in out


@pytest.mark.parametrize("np_", [1, 2, 4])
def test_para2d_spmm_driver(np_):
    out = _run("test_para2d_spmm.exe", np_, "g_symm.mtx", 64)
    assert "2D process grid: pm, pn =" in out and "Total para2d_spmm_exec()" in out
    assert "Replicate A matrix (once)" in out


def test_rp_spmm_driver_env_knobs():
    out = _run("test_rp_spmm.exe", 2, "g_symm.mtx", 16, {"RP_SPMM_P2P": "0", "RP_SPMM_REIDX": "0"})
    assert "Overriding parameter rB_reidx: 1 (default) --> 0 (runtime)" in out


@pytest.mark.parametrize("np_", [1, 2, 4])
def test_crpspmm_engine_driver(np_):
    """the older all-in-one API (deprecated/examples/test_crpspmm.c): <ntest> <check> <use-CUDA>"""
    out = _run("test_crpspmm.exe", np_, "g_symm.mtx", 48, tail=("2", "1", "1"))
    assert "CRP-SpMM 2D partition:" in out and "Redist C to user's 2D layout" in out
    assert "Alltoallv B necessary" in out
    out = _run("test_crpspmm.exe", np_, "g_gen.mtx", 5, tail=("1", "1"))      # non-square A, narrow B
    assert "SpMM total (avg of   1 runs)" in out


def test_crpspmm_engine_driver_env_knob():
    out = _run("test_crpspmm.exe", 2, "g_symm.mtx", 16, {"A2A_B_FINEGRAIN": "1"}, tail=("1", "1"))
    assert "Overriding parameter a2a_B_finegrain: 0 (default) --> 1 (runtime)" in out


@pytest.mark.parametrize("np_", [1, 2, 3])
def test_mpi_backend_device_exchange(np_):
    """include/crp_mpi.h: the device all-to-all of the MPI communicator back end on its own.  One rank
    has the GPU to itself and goes through RCCL (grouped ncclSend / ncclRecv); several ranks on the one
    GPU of the test box fall back to host staging -- both must deliver every block."""
    path = os.path.join(ROOT, "examples", "rccl_probe.exe")
    if not os.path.exists(path) or not os.path.exists(MPIEXEC):
        pytest.skip("no MPI launcher / example programs not built on this machine")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env["PATH"] = os.path.dirname(MPIEXEC) + ":" + env["PATH"]
    r = subprocess.run([MPIEXEC, "-np", str(np_), path], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ("transport rccl, ok" if np_ == 1 else "transport host-staged, ok") in r.stdout, r.stdout[-500:]
