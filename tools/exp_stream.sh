mkdir -p gpurun_out/r4_g21
EXP_DEBUG=1 timeout -k 10 300 python tools/exp_ctx.py 2>&1 | grep -v "query [0-9]* m=\|range [0-9]*: T=\|host time of a timed\|pipeline launch\|plan candidate\|bulk on\|workgroups, " > gpurun_out/r4_g21/ctx.txt
echo done
