set -x
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench44.json 2> gpurun_out/r3_bench44.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench44.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['value_incl_h2d_first'], d['bit_exact_vs_reference'])
for r in d['secondary']: print(r['workload'], r['value'], r['value_incl_h2d'], r['value_incl_h2d_first'], r['bit_exact_vs_reference'])
"
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k contract 2>&1 | tail -n 3
