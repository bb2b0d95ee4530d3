set -x
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "score_profile" > gpurun_out/r3_t11a.log 2>&1; echo "sp parity rc=$?"; tail -n 4 gpurun_out/r3_t11a.log
python -m pytest tests/test_cli.py -m gpu -x -q > gpurun_out/r3_t11b.log 2>&1; echo "cli rc=$?"; tail -n 4 gpurun_out/r3_t11b.log
python tools/ab_kernels.py --scale 0.3 --rounds 3 "query_profile:" "score_profile:sp_threshold=0" > gpurun_out/r3_ab_profile_technique.txt 2>&1; tail -n 3 gpurun_out/r3_ab_profile_technique.txt
python tools/short_query_bench.py > gpurun_out/r3_sq_qmajor.log 2>&1; head -3 gpurun_out/r3_sq_qmajor.log
SWIMM_HIP_OPTIONS=batch_order=0 python tools/short_query_bench.py > gpurun_out/r3_sq_gmajor.log 2>&1; head -3 gpurun_out/r3_sq_gmajor.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r3_t11c.log 2>&1; echo "fuzz+edges rc=$?"; tail -n 3 gpurun_out/r3_t11c.log
