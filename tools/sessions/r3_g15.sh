set -x
SWIMM_FUZZ_FIRST=2000 SWIMM_FUZZ_SEEDS=600 SWIMM_FUZZ_SESSIONS=0 SWIMM_FUZZ_SHORT=0 timeout -k 10 700 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "random_case" > gpurun_out/r3_t15.log 2>&1; echo "fuzz 2000.. rc=$?"; tail -n 4 gpurun_out/r3_t15.log
SWIMM_BENCH_SHARE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.05 > gpurun_out/r3_bench15_torchrun.json 2> gpurun_out/r3_bench15_torchrun.err; echo "torchrun bench rc=$?"; tail -c 600 gpurun_out/r3_bench15_torchrun.json
python bench.py --workload c5 --scale 1 --steps 2 --warmup 1 > gpurun_out/r3_bench15_c5_full.json 2> gpurun_out/r3_bench15_c5_full.err; echo "c5 full rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r3_bench15_c5_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_incl_h2d'], d['bit_exact_vs_reference'], d['parity_sample'], d['valu_roofline']['frac'], d['valu_roofline']['instructions'][:80])
"
