#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_precision.sh <tag>
# rocprofv3 kernel-trace statistics of the cfg5-shard log-L kernel in fp64 / mixed / fp32 and two SQ PMC passes of the same
# program, each in its own run (counters never share a run with trace domains other than --kernel-trace).
set -e -o pipefail
TAG=${1:-profprec}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/scripts/profile_precision.py"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $P > $OUT/stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmcA -- $P --light > $OUT/pmcA.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcB -- $P --light > $OUT/pmcB.log 2>&1
cd $R
cat $OUT/stats/*/*_kernel_stats.csv > $OUT/kernel_stats.csv
python3 scripts/pmc_summary.py "$OUT/pmc*/*/*_counter_collection.csv" > $OUT/pmc_summary.txt
cat $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt
grep -h "cfg5 shard" $OUT/stats.log
