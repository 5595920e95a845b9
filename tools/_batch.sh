set -o pipefail
for ro in auto 1 0; do
  if [ $ro = auto ]; then unset CRPSPMM_REORDER; else export CRPSPMM_REORDER=$ro; fi
  for cfg in "kkt 256" "fem3d 256"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --steps 30 --no-cpu-baseline --no-also --matrix $1 --n $2 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('reorder $ro $1 n=$2:', round(d['ms_per_step'],4), 'frac %.3f'%d['roofline']['frac'], 'reordered', d['config'].get('locality_order'), 'lattice', d['config'].get('lattice_detected'), 'first_exec', round(d['config']['first_exec_s'],2))"
  done
done
