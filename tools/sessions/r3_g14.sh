set -x
SWIMM_FUZZ_SEEDS=420 SWIMM_FUZZ_SESSIONS=90 SWIMM_FUZZ_SHORT=50 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t14.log 2>&1; echo "wide fuzz rc=$?"; tail -n 6 gpurun_out/r3_t14.log
