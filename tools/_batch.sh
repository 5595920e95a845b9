set -o pipefail
mkdir -p gpurun_out/b15
./tools/isa_probe > gpurun_out/b15/isa_probe.txt 2>&1 || exit 1
tail -8 gpurun_out/b15/isa_probe.txt
for v in 0 1; do
timeout -k 10 300 python bench.py --dtype f32 --matrix fem3d --n 1024 --steps 30 --variant $v --no-cpu-baseline > gpurun_out/b15/bench_f32_fem3d_v$v.json 2> gpurun_out/b15/bench_f32_v$v.err || { tail -5 gpurun_out/b15/bench_f32_v$v.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/b15/bench_f32_fem3d_v$v.json'));print('f32 fem3d n=1024 variant $v:', d['roofline']['kernel'][-70:], 'frac %.3f'%d['roofline']['frac'], 'GFLOP/s %.0f'%d['value'], d['config']['check_rel_err'])"
done
timeout -k 10 300 python bench.py --matrix fem3d --n 1024 --steps 30 --no-cpu-baseline > gpurun_out/b15/bench_f64_fem3d.json 2> gpurun_out/b15/bench_f64.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/b15/bench_f64_fem3d.json'));print('f64 fem3d n=1024:', d['roofline']['kernel'][-60:], 'frac %.3f'%d['roofline']['frac'], 'GFLOP/s %.0f'%d['value'])"
