python tools/short_query_bench.py 2>&1 | tee gpurun_out/r3_sq_resfactor.log
python tools/bench_configs.py --config c5 --scale 0.02 | tail -n 1
python tools/bench_configs.py --config c3 --scale 0.1 | tail -n 1
