set -e
for opt in "" "--max-waves 8" "--max-waves 12" "--rows-per-wave 16" "--rows-per-wave 32" "--rows-per-wave 24"; do
  echo "== $opt"
  python tools/bench_configs.py --config c5 --scale 0.06 --per-query --only 0,1,2,3,4,5,6,7,8,9 --reps 1 $opt | grep query_len
done
