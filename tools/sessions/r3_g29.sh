set -x
for o in "resident=-1" "resident=1" "resident=1 --opt rows_per_wave=24 --opt waves=4" "resident=1 --opt rows_per_wave=24 --opt waves=8" "resident=0 --opt rows_per_wave=24 --opt waves=8" "resident=0 --opt rows_per_wave=24 --opt waves=16"; do
  echo "c2 $o: $(python tools/bench_configs.py --config c2 --scale 1.0 --reps 4 --opt $o 2>&1 | grep -i gcups | tail -n 1)"
done
for o in "resident=-1" "resident=1"; do
  echo "c4@0.186 $o: $(python tools/bench_configs.py --config c4 --scale 0.186 --opt $o 2>&1 | grep -i gcups | tail -n 1)"
done
