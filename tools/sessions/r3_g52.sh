set -x
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench52.json 2> gpurun_out/r3_bench52.err; echo "bench rc=$?"; wc -c gpurun_out/r3_bench52.json
SWIMM_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.02 > gpurun_out/r3_bench52_2ranks.json 2> gpurun_out/r3_bench52_2ranks.err; echo "bench2 rc=$?"; wc -c gpurun_out/r3_bench52_2ranks.json
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k contract 2>&1 | tail -n 2
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench52.json').read().strip().splitlines()[-1])
print(d['value'], d['value_incl_h2d'], d['value_incl_h2d_first'], [ (r['workload'], r['value'], r['roofline'].get('lds')) for r in d['secondary']])
"
