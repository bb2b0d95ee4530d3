set -x
for lib in libswimm_hip.so libswimm_hip_lb32.so; do
  echo "== $lib"
  for R in 1 0; do echo "T=32 waves=4 resident=$R: $(SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=$R --opt waves=4 --opt rows_per_wave=32 2>&1 | grep -i gcups | tail -n 1 | cut -c1-120)"; done
  echo "T=32 waves=8 resident=1: $(SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/bench_configs.py --config c5 --scale 0.05 --opt resident=1 --opt waves=8 --opt rows_per_wave=32 2>&1 | grep -i gcups | tail -n 1 | cut -c1-120)"
  echo "T=32 waves=16 resident=0 (c2 forced): $(SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/bench_configs.py --config c2 --scale 1.0 --opt resident=0 --opt waves=16 --opt rows_per_wave=32 2>&1 | grep -i gcups | tail -n 1 | cut -c1-120)"
done
