set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r3_t1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
SWIMM_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.02 > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; echo "bench2 rc=$?"
SWIMM_HIP_DEBUG=1 python tools/bench_configs.py --config c3 --scale 0.1 --reps 1 > gpurun_out/r3_c3_10_dbg.log 2>&1
python tools/bench_configs.py --config c3 --scale 0.1 > gpurun_out/r3_c3_10.log 2>&1
python tools/bench_configs.py --config c3 --scale 0.3 > gpurun_out/r3_c3_30.log 2>&1
python tools/bench_configs.py --config c3 --scale 1.0 > gpurun_out/r3_c3_100.log 2>&1
tail -2 gpurun_out/r3_c3_10.log gpurun_out/r3_c3_30.log gpurun_out/r3_c3_100.log
