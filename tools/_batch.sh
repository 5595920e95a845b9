set -o pipefail
mkdir -p gpurun_out/b13
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/b13/pytest.txt 2>&1; rc=$?; tail -15 gpurun_out/b13/pytest.txt
exit $rc
