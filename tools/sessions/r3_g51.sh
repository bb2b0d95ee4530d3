set -x
bash tools/profile_bench.sh r3c4c --workload c4
bash tools/profile_bench.sh r3c5c --workload c5 --steps 1 --warmup 0
