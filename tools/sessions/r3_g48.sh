set -x
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py -m gpu -x -q -k "every_residue_code" > gpurun_out/r3_t48.log 2>&1; echo "rc=$?"; tail -n 15 gpurun_out/r3_t48.log
