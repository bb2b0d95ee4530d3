set -x
for sc in 0.1 0.3; do for o in "tall=0" "tall=1" "tall=-1"; do
  echo "c3@$sc $o: $(python tools/bench_configs.py --config c3 --scale $sc --opt $o 2>&1 | grep -i gcups | tail -n 1 | cut -c1-150)"
done; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench43.json 2> gpurun_out/r3_bench43.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench43.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['valu_roofline']['class_frac'], d['valu_roofline']['class_measured_frac'], d['roofline']['lds'])
"
