set -x
for sc in 0.1 0.3 1.0; do python tools/bench_configs.py --config c3 --scale $sc --check 40 > gpurun_out/r3_c3_${sc}_k.log 2>&1; tail -n 2 gpurun_out/r3_c3_${sc}_k.log; done
for k in 2 3 4; do for sc in 0.1 0.3 1.0; do echo "bulk_streams $k scale $sc"; python tools/bench_configs.py --config c3 --scale $sc --opt bulk_streams=$k | tail -n 1; done; done
SWIMM_HIP_DEBUG=1 python tools/bench_configs.py --config c3 --scale 0.3 --reps 1 2>&1 | grep "makespan" | head -2
python tools/bench_configs.py --config c5 --scale 0.02 --check 40 | tail -n 2
python -m pytest tests/test_gpu_edges.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; echo "edges+configs rc=$?"; tail -n 3 gpurun_out/r3_t8.log
python tools/preprocess_scale.py 1.0e9 gpurun_out/r3_preprocess_scale.json
