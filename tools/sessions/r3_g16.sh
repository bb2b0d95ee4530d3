set -x
bash tools/profile_cmd.sh r3res8 tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=1 --opt waves=8 --opt rows_per_wave=24
bash tools/profile_cmd.sh r3pp8 tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=0 --opt waves=8 --opt rows_per_wave=24
bash tools/profile_cmd.sh r3res4 tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=1
