set -x
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench47.json 2> gpurun_out/r3_bench47.err; echo "bench rc=$?"
SWIMM_BENCH_SHARE_DEVICE=1 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.02 > gpurun_out/r3_bench47_2ranks.json 2> gpurun_out/r3_bench47_2ranks.err; echo "bench2 rc=$?"
SWIMM_BENCH_SHARE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.1 --strong-scale 0.05 > gpurun_out/r3_bench47_torchrun.json 2> gpurun_out/r3_bench47_torchrun.err; echo "torchrun bench rc=$?"
python -c "
import json
for f in ('gpurun_out/r3_bench47.json','gpurun_out/r3_bench47_2ranks.json','gpurun_out/r3_bench47_torchrun.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['n_gpus'], d.get('value_incl_h2d'), d.get('bit_exact_vs_reference'), d['strong_scaling']['value'], d['strong_scaling'].get('bit_exact'), d.get('dtype'))
"
