set -x
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t25.log 2>&1; rc=$?; echo "parity/edges/fuzz rc=$rc"; tail -n 25 gpurun_out/r3_t25.log
python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r3_bench25.json 2> gpurun_out/r3_bench25.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench25.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'])
"
