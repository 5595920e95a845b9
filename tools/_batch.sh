set -o pipefail
mkdir -p gpurun_out/kkt240
export CRPSPMM_CACHE_DIR=/tmp
for n in 256 128; do
  timeout -k 10 1000 python bench.py --matrix kkt240 --n $n --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/kkt240/bench_n$n.json 2> gpurun_out/kkt240/bench_n$n.err; rc=$?
  tail -12 gpurun_out/kkt240/bench_n$n.err | grep -v amdgpu.ids
  [ $rc -eq 0 ] || exit 1
  cut -c1-900 gpurun_out/kkt240/bench_n$n.json
done
