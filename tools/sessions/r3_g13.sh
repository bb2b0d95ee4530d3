set -x
for pt in 0 1; do
 echo "== prio_top=$pt"
 python tools/bench_configs.py --config c5 --scale 0.05 --opt prio_top=$pt --opt resident=1 --opt waves=8 --opt rows_per_wave=24 | tail -n 1
 python tools/bench_configs.py --config c5 --scale 0.05 --opt prio_top=$pt --opt resident=1 | tail -n 1
 python tools/bench_configs.py --config c5 --scale 0.05 --opt prio_top=$pt | tail -n 1
 python tools/bench_configs.py --config c3 --scale 0.1 --opt prio_top=$pt | tail -n 1
 python tools/bench_configs.py --config c3 --scale 0.1 --opt prio_top=$pt --opt tall=1 | tail -n 1
 python tools/bench_configs.py --config c3 --scale 1.0 --opt prio_top=$pt | tail -n 1
 SQ_ONLY=0,2 SWIMM_HIP_OPTIONS=prio_top=$pt python tools/short_query_bench.py
done
python tools/ab_kernels.py --scale 1.0 --rounds 5 "base:" "prio_top:prio_top=1"
