set -x
SWIMM_FUZZ_GE_MAX=45 SWIMM_FUZZ_FIRST=2000 SWIMM_FUZZ_SEEDS=500 SWIMM_FUZZ_SESSIONS=40 SWIMM_FUZZ_SHORT=20 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t54.log 2>&1; echo "fuzz with extend penalties up to 44 rc=$?"; tail -n 4 gpurun_out/r3_t54.log
