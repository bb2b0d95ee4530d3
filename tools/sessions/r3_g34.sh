set -x
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t34.log 2>&1; rc=$?; echo "parity/edges/fuzz rc=$rc"; tail -n 4 gpurun_out/r3_t34.log
python tools/plan_sweep.py --scale 1.0 --ws 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 > gpurun_out/r3_plan_sweep2.txt 2>&1; echo "sweep rc=$?"; tail -n 2 gpurun_out/r3_plan_sweep2.txt
