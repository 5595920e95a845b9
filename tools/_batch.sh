set -o pipefail
for n in 64 128 192 256; do
  for v in 1 5; do
    timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline --no-also --matrix fem3d --n $n --dtype f32 --variant $v 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('fem3d f32 n=$n variant $v:', round(d['ms_per_step'],4), 'ms')"
  done
done
