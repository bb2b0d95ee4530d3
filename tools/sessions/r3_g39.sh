set -x
for i in 1 2; do
for lib in libswimm_hip.so libswimm_hip_b64.so; do
  echo "== $lib"
  SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], d['bit_exact_vs_reference'], d['config']['plan'])"
done; done
for lib in libswimm_hip.so libswimm_hip_b64.so; do
  echo "== $lib"
  SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/plan_sweep.py --scale 1.0 --ws 4,8,12,16 2>&1 | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
for W in (4,8,12,16): print('W=%d'%W, ' '.join('%d:%d'%(r['T'],r['gcups']) for r in rows if r['W']==W))"
  SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/bench_configs.py --config c5 --scale 0.05 2>&1 | grep -i gcups | tail -n 1
done
