set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t24.log 2>&1; echo "parity/edges/fuzz rc=$?"; tail -n 3 gpurun_out/r3_t24.log
{
for T in 16 20 24 28 32 36; do for R in 1 0; do
  echo "T=$T waves=4 resident=$R: $(python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=$R --opt waves=4 --opt rows_per_wave=$T 2>&1 | grep -i "gcups" | tail -n 1)"
done; done
for T in 24 28; do for R in 1 0; do
  echo "T=$T waves=8 resident=$R: $(python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=$R --opt waves=8 --opt rows_per_wave=$T 2>&1 | grep -i "gcups" | tail -n 1)"
done; done
echo "T=28 waves=16 resident=1: $(python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=1 --opt waves=16 --opt rows_per_wave=28 2>&1 | grep -i "gcups" | tail -n 1)"
} > gpurun_out/r3_res24.log 2>&1
cat gpurun_out/r3_res24.log
python tools/short_query_bench.py > gpurun_out/r3_sq24_017.log 2>&1; cat gpurun_out/r3_sq24_017.log
SQ_SCALE=0.4 python tools/short_query_bench.py > gpurun_out/r3_sq24_04.log 2>&1; cat gpurun_out/r3_sq24_04.log
SQ_SCALE=1.0 python tools/short_query_bench.py > gpurun_out/r3_sq24_10.log 2>&1; cat gpurun_out/r3_sq24_10.log
python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/r3_bench24.json 2> gpurun_out/r3_bench24.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench24.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'])
for r in d['secondary']: print(r['workload'], r['value'], r['bit_exact_vs_reference'], r['value_incl_h2d'])
"
