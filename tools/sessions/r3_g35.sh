set -x
{
for T in 16 20 24 28 32 36; do for R in 1 0; do
  echo "T=$T waves=4 resident=$R: $(python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=$R --opt waves=4 --opt rows_per_wave=$T 2>&1 | grep -i "gcups" | tail -n 1)"
done; done
for T in 24 32; do for R in 1 0; do
  echo "T=$T waves=8 resident=$R: $(python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=$R --opt waves=8 --opt rows_per_wave=$T 2>&1 | grep -i "gcups" | tail -n 1)"
done; done
} > gpurun_out/r3_res35.log 2>&1
cat gpurun_out/r3_res35.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench35.json 2> gpurun_out/r3_bench35.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench35.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['config']['plan'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'], r['config']['plan'])
"
for sc in 1.0 0.3 0.1; do echo "c3 scale $sc: $(python tools/bench_configs.py --config c3 --scale $sc 2>&1 | grep -i gcups | tail -n 1)"; done
python tools/short_query_bench.py
