set -x
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t30.log 2>&1; rc=$?; echo "parity/edges/fuzz rc=$rc"; tail -n 4 gpurun_out/r3_t30.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench30.json 2> gpurun_out/r3_bench30.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench30.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['config']['plan'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'], r['config']['plan'])
"
