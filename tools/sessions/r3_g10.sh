set -x
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score_profile or golden_all or interleaved" > gpurun_out/r3_t10a.log 2>&1; echo "sp parity rc=$?"; tail -n 4 gpurun_out/r3_t10a.log
python -m pytest tests/test_cli.py -m gpu -x -q > gpurun_out/r3_t10b.log 2>&1; echo "cli rc=$?"; tail -n 4 gpurun_out/r3_t10b.log
python tools/ab_kernels.py --scale 0.3 --rounds 3 "query_profile:" "score_profile:sp_threshold=0" > gpurun_out/r3_ab_profile_technique.txt 2>&1; cat gpurun_out/r3_ab_profile_technique.txt | tail -n 3
bash tools/profile_bench.sh r3c2
bash tools/profile_bench.sh r3c4 --workload c4
bash tools/profile_bench.sh r3c5 --workload c5 --steps 1 --warmup 0
python tools/short_query_bench.py > gpurun_out/r3_sq_final_017.log 2>&1
SQ_SCALE=0.4 python tools/short_query_bench.py > gpurun_out/r3_sq_final_04.log 2>&1
SQ_SCALE=1.0 python tools/short_query_bench.py > gpurun_out/r3_sq_final_10.log 2>&1
python tools/upload_timeline.py chunks > gpurun_out/r3_upload_timeline_chunks3.txt 2>&1; grep "^rep" gpurun_out/r3_upload_timeline_chunks3.txt
python tools/upload_timeline.py slabs > gpurun_out/r3_upload_timeline_slabs3.txt 2>&1; grep "^rep" gpurun_out/r3_upload_timeline_slabs3.txt
