set -x
python -m pytest tests/test_gpu_edges.py -m gpu -x -q -k "outlier or streamed or chained or length" > gpurun_out/r3_t5a.log 2>&1; echo "edges rc=$?"; tail -n 4 gpurun_out/r3_t5a.log
for sc in 0.1 0.3 1.0; do python tools/bench_configs.py --config c3 --scale $sc --check 60 > gpurun_out/r3_c3_${sc}_cut.log 2>&1; tail -n 2 gpurun_out/r3_c3_${sc}_cut.log; done
for cut in 0 15 60; do python tools/bench_configs.py --config c3 --scale 1.0 --opt cut=$cut > gpurun_out/r3_c3_1.0_cut$cut.log 2>&1; tail -n 1 gpurun_out/r3_c3_1.0_cut$cut.log; done
python tools/bench_configs.py --config c3 --scale 0.3 --opt cut=0 | tail -n 1
python tools/upload_timeline.py chunks > gpurun_out/r3_upload_timeline_chunks2.txt 2>&1; grep "^rep\|pipeline launch" gpurun_out/r3_upload_timeline_chunks2.txt
python tools/upload_timeline.py chunks 1.0 upload_head=0 > gpurun_out/r3_upload_timeline_chunks_nohead.txt 2>&1; grep "^rep" gpurun_out/r3_upload_timeline_chunks_nohead.txt
python tools/upload_timeline.py slabs > gpurun_out/r3_upload_timeline_slabs2.txt 2>&1; grep "^rep" gpurun_out/r3_upload_timeline_slabs2.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r3_t5b.log 2>&1; echo "parity+fuzz+configs rc=$?"; tail -n 4 gpurun_out/r3_t5b.log
