mkdir -p gpurun_out/r4_g14
timeout -k 10 200 python tools/cold_timeline.py c4 0.186 > gpurun_out/r4_g14/cold_c4.txt 2>&1
timeout -k 10 60 python tools/upload_timeline.py chunks 1.0 2>&1 | grep -v "timed launch\|amdgpu\|candidate" > gpurun_out/r4_g14/t_chunks.txt
timeout -k 10 100 python tools/cold_timeline.py c3 1.0 > gpurun_out/r4_g14/cold_c3.txt 2>&1
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r4_g14/tests.txt 2>&1
echo done
