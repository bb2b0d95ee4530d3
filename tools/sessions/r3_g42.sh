set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t42.log 2>&1; echo "full gpu suite rc=$?"; tail -n 5 gpurun_out/r3_t42.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench42.json 2> gpurun_out/r3_bench42.err; echo "bench rc=$?"
SWIMM_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.02 > gpurun_out/r3_bench42_2ranks.json 2> gpurun_out/r3_bench42_2ranks.err; echo "bench2 rc=$?"
bash tools/profile_bench.sh r3c2c
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench42.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['roofline'], d['valu_roofline'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'], r['roofline']['traffic'], r['valu_roofline']['instructions'][:60])
print(d['strong_scaling']['value'], d['strong_scaling']['hbm_frac'])
"
