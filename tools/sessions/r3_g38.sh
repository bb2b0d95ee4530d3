set -x
bash tools/profile_bench.sh r3c4b --workload c4
bash tools/profile_bench.sh r3c5b --workload c5 --steps 1 --warmup 0
python tools/short_query_bench.py > gpurun_out/r3_sq38_017.log 2>&1
SQ_SCALE=0.4 python tools/short_query_bench.py > gpurun_out/r3_sq38_04.log 2>&1
SQ_SCALE=1.0 python tools/short_query_bench.py > gpurun_out/r3_sq38_10.log 2>&1
for sc in 1.0 0.3 0.1; do echo "c3 scale $sc: $(python tools/bench_configs.py --config c3 --scale $sc 2>&1 | grep -i gcups | tail -n 1)"; done > gpurun_out/r3_c3_38.txt 2>&1
python tools/upload_timeline.py chunks > gpurun_out/r3_upload_timeline_chunks38.txt 2>&1; grep "^rep" gpurun_out/r3_upload_timeline_chunks38.txt
python tools/upload_timeline.py slabs > gpurun_out/r3_upload_timeline_slabs38.txt 2>&1; grep "^rep" gpurun_out/r3_upload_timeline_slabs38.txt
