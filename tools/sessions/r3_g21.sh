set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t21.log 2>&1; echo "full gpu suite rc=$?"; tail -n 5 gpurun_out/r3_t21.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench21.json 2> gpurun_out/r3_bench21.err; echo "bench rc=$?"
SWIMM_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.02 > gpurun_out/r3_bench21_2ranks.json 2> gpurun_out/r3_bench21_2ranks.err; echo "bench2 rc=$?"
python tools/short_query_bench.py > gpurun_out/r3_sq_final2_017.log 2>&1
SQ_SCALE=0.4 python tools/short_query_bench.py > gpurun_out/r3_sq_final2_04.log 2>&1
SQ_SCALE=1.0 python tools/short_query_bench.py > gpurun_out/r3_sq_final2_10.log 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench21.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['roofline']['traffic_source'], d['roofline'].get('lds'))
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'], r['roofline']['traffic'], r['valu_roofline']['instructions'][:60])
print(d['strong_scaling']['value'], d['strong_scaling']['hbm_frac'])
"
