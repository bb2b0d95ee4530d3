set -x
export SWIMM_HIP_LIB=$PWD/swimm_amd/lib/libswimm_hip_stamps.so
for cfg in "resident=1 rows_per_wave=24 waves=8" "resident=0 rows_per_wave=24 waves=8" "resident=1 rows_per_wave=32 waves=4" "resident=0 rows_per_wave=32 waves=4"; do
  tag=$(echo $cfg | tr ' =' '__')
  opts=""; for kv in $cfg; do opts="$opts --opt $kv"; done
  python tools/bench_configs.py --config c5 --scale 0.02 --only 19 --reps 1 $opts > gpurun_out/r3_stamps_c5_$tag.txt 2>&1
  grep "stamps" gpurun_out/r3_stamps_c5_$tag.txt | tail -n 12; tail -n 1 gpurun_out/r3_stamps_c5_$tag.txt
done
SQ_ONLY=0 python tools/short_query_bench.py > gpurun_out/r3_stamps_sq_res.txt 2>&1; grep "stamps" gpurun_out/r3_stamps_sq_res.txt | tail -n 8; tail -n 1 gpurun_out/r3_stamps_sq_res.txt
SQ_ONLY=0 SWIMM_HIP_OPTIONS=resident=0,rotate=0 python tools/short_query_bench.py > gpurun_out/r3_stamps_sq_pp.txt 2>&1; grep "stamps" gpurun_out/r3_stamps_sq_pp.txt | tail -n 8; tail -n 1 gpurun_out/r3_stamps_sq_pp.txt
