set -x
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t40.log 2>&1; rc=$?; echo "parity/edges/fuzz rc=$rc"; tail -n 4 gpurun_out/r3_t40.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench40.json 2> gpurun_out/r3_bench40.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench40.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['config']['plan'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'], r['config']['plan'])
"
python tools/plan_sweep.py --scale 1.0 --ws 4,8,16 2>&1 | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
for W in (4,8,16): print('W=%d'%W, ' '.join('%d:%d'%(r['T'],r['gcups']) for r in rows if r['W']==W))"
