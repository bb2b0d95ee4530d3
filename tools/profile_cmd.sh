#!/bin/bash
# Profiles any python command of this repository on the GPU box: kernel trace + stats, then the HBM-traffic and
# instruction counters, each PMC pass in its own run (MI355X_MICROARCH.md: --pmc never combined with a trace domain).
# usage: tools/profile_cmd.sh <tag> <script.py> [args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 "$@" > $OUT/kt.log 2>&1
echo "kt rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -o pmc -- python3 "$@" > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc -- python3 "$@" > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -o pmc -- python3 "$@" > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?"
