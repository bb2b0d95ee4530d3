set -x
SWIMM_FUZZ_FIRST=160 SWIMM_FUZZ_SEEDS=900 SWIMM_FUZZ_SESSIONS=160 SWIMM_FUZZ_SHORT=60 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t45.log 2>&1; echo "extended fuzz rc=$?"; tail -n 4 gpurun_out/r3_t45.log
