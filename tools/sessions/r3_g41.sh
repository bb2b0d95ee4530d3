set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r3lds
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_r3lds/pmc2 -o pmc -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cold --no-secondary > gpurun_out/prof_r3lds/pmc2.log 2>&1
echo "pmc2 rc=$?"
python - <<'P'
import csv,glob,collections
fn=glob.glob('gpurun_out/prof_r3lds/pmc2/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for r in csv.DictReader(open(fn)):
    if 'sw_pipe_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
for k,v in acc.items(): print(k, v/len(n[k]))
print('conflict frac', acc['SQ_LDS_BANK_CONFLICT']/acc['SQ_LDS_IDX_ACTIVE'])
P
