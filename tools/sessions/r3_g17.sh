set -x
export SWIMM_HIP_LIB=$PWD/swimm_amd/lib/libswimm_hip_stamps.so
python tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=1 --opt waves=8 --opt rows_per_wave=24 > gpurun_out/r3_stamps2_res8.txt 2>&1; grep "stamps" gpurun_out/r3_stamps2_res8.txt | tail -n 10
python tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=0 --opt waves=8 --opt rows_per_wave=24 > gpurun_out/r3_stamps2_pp8.txt 2>&1; grep "stamps wave" gpurun_out/r3_stamps2_pp8.txt | tail -n 8
python tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=1 --opt waves=4 --opt rows_per_wave=32 > gpurun_out/r3_stamps2_res4.txt 2>&1; grep "stamps" gpurun_out/r3_stamps2_res4.txt | tail -n 6
python tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=1 --opt waves=4 --opt rows_per_wave=24 > gpurun_out/r3_stamps2_res4x24.txt 2>&1; grep "stamps" gpurun_out/r3_stamps2_res4x24.txt | tail -n 6
python tools/bench_configs.py --config c5 --scale 0.05 --reps 1 --opt resident=0 --opt waves=4 --opt rows_per_wave=24 > gpurun_out/r3_stamps2_pp4x24.txt 2>&1; grep "stamps wave" gpurun_out/r3_stamps2_pp4x24.txt | tail -n 4
