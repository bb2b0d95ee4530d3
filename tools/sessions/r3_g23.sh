set -x
for o in "" "resident=1" "resident=0"; do
  echo "== 0.27 / 80-120, 20-60 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=0.27 SQ_ONLY=0,2 python tools/short_query_bench.py
done > gpurun_out/r3_sq23.log 2>&1
for o in "" "resident=1"; do
  echo "== 1.0 / 400-440 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=1.0 SQ_SET=100,400,440 python tools/short_query_bench.py
  echo "== 1.0 / 450-560 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=1.0 SQ_SET=100,450,560 python tools/short_query_bench.py
  echo "== 2.0 / 280-320 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=2.0 SQ_ONLY=1 python tools/short_query_bench.py
  echo "== 2.0 / 400-440 / $o"; SWIMM_HIP_OPTIONS="$o" SQ_SCALE=2.0 SQ_SET=60,400,440 python tools/short_query_bench.py
done >> gpurun_out/r3_sq23.log 2>&1
grep -v "^+" gpurun_out/r3_sq23.log
