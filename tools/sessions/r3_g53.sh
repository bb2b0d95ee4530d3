set -x
export SWIMM_HIP_LIB=$PWD/swimm_amd/lib/libswimm_hip_stamps.so
export SWIMM_HIP_DEBUG=1
for opts in "" "--opt rows_per_wave=32 --opt waves=4" "--opt rows_per_wave=24 --opt waves=4 --opt resident=1"; do
  tag=$(echo "$opts" | tr -d ' =-' | cut -c1-30)
  python tools/bench_configs.py --config c2 --scale 1.0 --reps 1 $opts > gpurun_out/r3_stamps53_$tag.txt 2>&1
  echo "== c2 $opts"; grep "stamps" gpurun_out/r3_stamps53_$tag.txt | tail -n 10 | cut -c1-260; tail -n 1 gpurun_out/r3_stamps53_$tag.txt | cut -c1-120
done
