set -x
python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3_t7a.log 2>&1; echo "edges+parity rc=$?"; tail -n 3 gpurun_out/r3_t7a.log
for sc in 0.1 0.3 1.0; do python tools/bench_configs.py --config c3 --scale $sc --check 60 > gpurun_out/r3_c3_${sc}_lad.log 2>&1; tail -n 2 gpurun_out/r3_c3_${sc}_lad.log; done
for cap in 5 12 40; do for frac in 30 60; do echo "cap $cap frac $frac"; python tools/bench_configs.py --config c3 --scale 1.0 --opt tail_cap=$cap --opt tail_frac=$frac | tail -n 1; done; done
for cap in 10 40 150; do echo "10%: cap $cap"; python tools/bench_configs.py --config c3 --scale 0.1 --opt tail_cap=$cap | tail -n 1; done
python tools/bench_configs.py --config c5 --scale 0.05 --check 40 | tail -n 2
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r3_t7b.log 2>&1; echo "fuzz+configs rc=$?"; tail -n 3 gpurun_out/r3_t7b.log
