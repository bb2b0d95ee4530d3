#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run, as
# MI355X_MICROARCH.md prescribes: --pmc never combined with other trace domains).
# usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-cold --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py $ARGS > $OUT/kt.log 2>&1
echo "kt rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc2 -o pmc -- python3 bench.py $ARGS > $OUT/pmc2.log 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -o pmc -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?"
find $OUT -name "*.csv" | head -30
