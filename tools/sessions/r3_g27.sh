set -x
python tools/plan_sweep.py --scale 1.0 --ws 1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16 > gpurun_out/r3_plan_sweep.txt 2>&1; echo "sweep rc=$?"; tail -n 3 gpurun_out/r3_plan_sweep.txt
for sc in 1.0 0.3 0.1; do echo "c3 scale $sc: $(python tools/bench_configs.py --config c3 --scale $sc 2>&1 | grep -i gcups | tail -n 1)"; done > gpurun_out/r3_c3_27.txt 2>&1; cat gpurun_out/r3_c3_27.txt
