set -x
SWIMM_HIP_DEBUG=1 python tools/bench_configs.py --config c3 --scale 1.0 --reps 1 2>&1 | grep -E "query [0-9]+ m=|tail of|gcups" | head -30
SWIMM_HIP_DEBUG=1 python tools/bench_configs.py --config c3 --scale 0.3 --reps 1 2>&1 | grep -E "query [0-9]+ m=|tail of|gcups" | head -30
