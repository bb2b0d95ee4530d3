set -x
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r3_t4a.log 2>&1; echo "parity+edges rc=$?"; tail -n 4 gpurun_out/r3_t4a.log
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_cli.py -m gpu -x -q > gpurun_out/r3_t4b.log 2>&1; echo "fuzz+configs rc=$?"; tail -n 4 gpurun_out/r3_t4b.log
for sc in 0.1 0.3; do python tools/bench_configs.py --config c3 --scale $sc --opt tall=1 > gpurun_out/r3_c3_${sc}_tall.log 2>&1; tail -n 1 gpurun_out/r3_c3_${sc}_tall.log; done
python tools/upload_timeline.py chunks > gpurun_out/r3_upload_timeline_chunks.txt 2>&1; head -3 gpurun_out/r3_upload_timeline_chunks.txt; tail -n 1 gpurun_out/r3_upload_timeline_chunks.txt
python tools/upload_timeline.py slabs > gpurun_out/r3_upload_timeline_slabs.txt 2>&1; tail -n 1 gpurun_out/r3_upload_timeline_slabs.txt
python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3_bench4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['value_incl_h2d_note'])
P
