#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_all.sh <tag>
# rocprofv3 kernel-trace statistics of every kernel (scripts/profile_all_kernels.py) and one SQ PMC pass of the same
# program in its own run (counters never share a run with trace domains other than --kernel-trace).
set -e -o pipefail
TAG=${1:-profall}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/scripts/profile_all_kernels.py > $OUT/stats.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmcA -- python3 $R/scripts/profile_all_kernels.py --light > $OUT/pmcA.log 2>&1
cd $R
cat $OUT/stats/*/*_kernel_stats.csv > $OUT/kernel_stats.csv
python3 scripts/pmc_summary.py "$OUT/pmcA/*/*_counter_collection.csv" > $OUT/pmc_summary.txt
cat $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt
