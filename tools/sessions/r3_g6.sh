set -x
python -m pytest tests/test_gpu_edges.py -m gpu -x -q -k "outlier" > gpurun_out/r3_t6a.log 2>&1; echo "edges rc=$?"; tail -n 3 gpurun_out/r3_t6a.log
for sc in 0.1 0.3 1.0; do python tools/bench_configs.py --config c3 --scale $sc --check 60 > gpurun_out/r3_c3_${sc}_cut2.log 2>&1; tail -n 2 gpurun_out/r3_c3_${sc}_cut2.log; done
for cut in 0 15 80; do python tools/bench_configs.py --config c3 --scale 1.0 --opt cut=$cut | tail -n 1; done
for cut in 0 15 80; do python tools/bench_configs.py --config c3 --scale 0.1 --opt cut=$cut | tail -n 1; done
bash tools/profile_cmd.sh r3c3_10 tools/bench_configs.py --config c3 --scale 0.1 --reps 1
bash tools/profile_cmd.sh r3c3_100 tools/bench_configs.py --config c3 --scale 1.0 --reps 1
python -m pytest tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3_t6b.log 2>&1; echo "configs+fuzz rc=$?"; tail -n 3 gpurun_out/r3_t6b.log
