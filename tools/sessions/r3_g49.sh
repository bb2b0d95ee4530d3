set -x
python tools/cold_twice.py 2> gpurun_out/r3_cold_twice.txt; grep -c "" gpurun_out/r3_cold_twice.txt
