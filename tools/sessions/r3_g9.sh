set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t9.log 2>&1; echo "full gpu suite rc=$?"; tail -n 6 gpurun_out/r3_t9.log
for sc in 0.1 0.3 1.0; do python tools/bench_configs.py --config c3 --scale $sc | tail -n 1; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench9.json 2> gpurun_out/r3_bench9.err; echo "bench rc=$?"
SWIMM_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.02 > gpurun_out/r3_bench9_2ranks.json 2> gpurun_out/r3_bench9_2ranks.err; echo "bench2 rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench9.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'])
"
