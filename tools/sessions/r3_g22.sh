set -x
python tools/upload_timeline.py > gpurun_out/r3_ut22.log 2>&1; tail -n 6 gpurun_out/r3_ut22.log
python bench.py --gpus 1 --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > gpurun_out/r3_bench22.json 2> gpurun_out/r3_bench22.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench22.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'])
"
for o in "" "resident=1" "resident=0"; do
  echo "== 1.0 / 280-320 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=1.0 SQ_ONLY=1,3 python tools/short_query_bench.py
  echo "== 0.17 / 80-120, 280-320 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=0.17 SQ_ONLY=0,1 python tools/short_query_bench.py
  echo "== 0.4 / 80-120, 20-60 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=0.4 SQ_ONLY=0,2 python tools/short_query_bench.py
done > gpurun_out/r3_sq22.log 2>&1
cat gpurun_out/r3_sq22.log
