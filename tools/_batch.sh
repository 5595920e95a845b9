set -o pipefail
mkdir -p gpurun_out/drv
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
make -C examples > gpurun_out/drv/make.log 2>&1 || { tail -5 gpurun_out/drv/make.log; exit 1; }
python tools/gen_mtx.py --kind banded_fem --m 217918 --out /tmp/pwtk_standin.mtx > gpurun_out/drv/gen.log 2>&1 || { tail -5 gpurun_out/drv/gen.log; exit 1; }
export PATH=/opt/conda/bin:$PATH
timeout -k 10 600 mpiexec -np 1 examples/test_rp_spmm.exe /tmp/pwtk_standin.mtx 256 5 0 1 > gpurun_out/drv/test_rp_spmm_np1.txt 2>&1 || { tail -20 gpurun_out/drv/test_rp_spmm_np1.txt; exit 1; }
tail -12 gpurun_out/drv/test_rp_spmm_np1.txt
timeout -k 10 600 mpiexec -np 2 examples/test_para2d_spmm.exe /tmp/pwtk_standin.mtx 256 5 0 1 > gpurun_out/drv/test_para2d_spmm_np2.txt 2>&1 || { tail -20 gpurun_out/drv/test_para2d_spmm_np2.txt; exit 1; }
tail -6 gpurun_out/drv/test_para2d_spmm_np2.txt
