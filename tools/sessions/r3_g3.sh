set -x
python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_t3a.log 2>&1; echo "edges+parity rc=$?"; tail -n 5 gpurun_out/r3_t3a.log
SWIMM_HIP_DEBUG=1 python tools/bench_configs.py --config c3 --scale 0.1 --reps 1 --check 100 > gpurun_out/r3_c3_10_dbg2.log 2>&1; tail -n 3 gpurun_out/r3_c3_10_dbg2.log
for sc in 0.1 0.3 1.0; do python tools/bench_configs.py --config c3 --scale $sc --check 60 > gpurun_out/r3_c3_${sc}_b.log 2>&1; tail -n 2 gpurun_out/r3_c3_${sc}_b.log; done
for sc in 0.1 0.3; do python tools/bench_configs.py --config c3 --scale $sc --opt tall=0 > gpurun_out/r3_c3_${sc}_notall.log 2>&1; tail -n 1 gpurun_out/r3_c3_${sc}_notall.log; done
python tools/bench_configs.py --config c3 --scale 1.0 --opt tall=1 > gpurun_out/r3_c3_1.0_tall.log 2>&1; tail -n 1 gpurun_out/r3_c3_1.0_tall.log
python tools/short_query_bench.py > gpurun_out/r3_sq_stack2.log 2>&1; cat gpurun_out/r3_sq_stack2.log
python -m pytest tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t3b.log 2>&1; echo "configs+fuzz rc=$?"; tail -n 5 gpurun_out/r3_t3b.log
