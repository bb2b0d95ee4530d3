set -x
python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench28.json 2> gpurun_out/r3_bench28.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench28.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['config'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'], r['config'])
"
SWIMM_DEBUG=1 python tools/bench_configs.py --config c2 --scale 1.0 2>&1 | grep -i "query 0 m=\|gcups" | tail -n 3
for sc in 1.0 0.3 0.1; do echo "c3 scale $sc: $(python tools/bench_configs.py --config c3 --scale $sc 2>&1 | grep -i gcups | tail -n 1)"; done
python tools/short_query_bench.py
SQ_SCALE=1.0 python tools/short_query_bench.py
