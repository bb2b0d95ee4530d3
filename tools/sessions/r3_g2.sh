set -x
python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "short_query" > gpurun_out/r3_t2a.log 2>&1; echo "short fuzz rc=$?"; tail -n 5 gpurun_out/r3_t2a.log
python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_cli.py -m gpu -x -q > gpurun_out/r3_t2b.log 2>&1; echo "edges rc=$?"; tail -n 5 gpurun_out/r3_t2b.log
python tools/short_query_bench.py > gpurun_out/r3_sq_stack.log 2>&1; cat gpurun_out/r3_sq_stack.log
SWIMM_HIP_OPTIONS=stack=0 python tools/short_query_bench.py > gpurun_out/r3_sq_nostack.log 2>&1; cat gpurun_out/r3_sq_nostack.log
SQ_SCALE=0.4 python tools/short_query_bench.py > gpurun_out/r3_sq_stack_04.log 2>&1; cat gpurun_out/r3_sq_stack_04.log
SQ_SCALE=1.0 python tools/short_query_bench.py > gpurun_out/r3_sq_stack_10.log 2>&1; cat gpurun_out/r3_sq_stack_10.log
