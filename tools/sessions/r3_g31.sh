set -x
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t31.log 2>&1; rc=$?; echo "parity/edges/fuzz rc=$rc"; tail -n 4 gpurun_out/r3_t31.log
