set -x
for i in 1 2; do
for lib in libswimm_hip.so libswimm_hip_asmmax.so; do
  SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['bit_exact_vs_reference'])"
done; done
for lib in libswimm_hip.so libswimm_hip_asmmax.so; do
  echo "== $lib"; SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/bench_configs.py --config c5 --scale 0.05 2>&1 | grep -i gcups | tail -n 1
  SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib python tools/bench_configs.py --config c4 --scale 0.1 2>&1 | grep -i gcups | tail -n 1
  SWIMM_HIP_LIB=$PWD/swimm_amd/lib/$lib SQ_ONLY=0,2,3 python tools/short_query_bench.py
done
