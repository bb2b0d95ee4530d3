mkdir -p gpurun_out/r4_g23
for sc in 0.1 0.3 1.0; do timeout -k 10 100 python tools/bench_configs.py --config c3 --scale $sc --reps 3 >> gpurun_out/r4_g23/c3_scales.txt 2>&1; done
timeout -k 10 700 python -m pytest tests -x -q -m gpu -k "c3 or edges or fuzz or golden or lane or tail" > gpurun_out/r4_g23/tests.txt 2>&1
echo done
