"""CPU: the host half of the product (format builders, team scheduler, planner, ingest) compiled with
AddressSanitizer + UBSan and driven by tests/host_asan.cpp (GPU sanitizers are not available on the pool)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_code_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "crp-spmm_amd", "csrc")
    files = [os.path.join(src, f) for f in ("panel_format.cpp", "team_order.cpp", "locality.cpp", "spmat_part.cpp", "mmio_utils.cpp", "host_support.cpp", "knobs.cpp")]
    exe = str(tmp_path / "host_asan")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + src,
           os.path.join(ROOT, "tests", "host_asan.cpp"), *files, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", CRPSPMM_NUM_THREADS="4", CRPSPMM_SYNC_RELEASE="1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "HOST_ASAN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
