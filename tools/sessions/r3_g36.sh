set -x
for cfg in "c2 1.0" "c3 0.3" "c3 1.0"; do set -- $cfg
for o in "resident=-1" "rows_per_wave=32 --opt waves=4" "rows_per_wave=32 --opt waves=4 --opt wgs_per_cu=3" "rows_per_wave=28 --opt waves=4" "rows_per_wave=28 --opt waves=4 --opt wgs_per_cu=3" "rows_per_wave=24 --opt waves=4" "rows_per_wave=24 --opt waves=4 --opt wgs_per_cu=3"; do
  echo "$1@$2 $o: $(python tools/bench_configs.py --config $1 --scale $2 --reps 3 --opt $o 2>&1 | grep -i gcups | tail -n 1 | cut -c1-140)"
done; done
