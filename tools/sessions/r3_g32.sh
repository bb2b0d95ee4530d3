set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_t32.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -n 4 gpurun_out/r3_t32.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r3_bench32.json 2> gpurun_out/r3_bench32.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['config']['plan'])
"
python tools/plan_sweep.py --scale 1.0 --ws 4,8,12,16 2>&1 | tail -n 33
