#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_gpu.sh <tag>
# Produces, under gpurun_out/<tag>/: kernel-trace stats of bench.py and four separate PMC passes
# (SQ counters x2, FETCH_SIZE, WRITE_SIZE — TCC cannot hold both in one pass), as the guide prescribes:
# counters are collected in their own runs with --kernel-trace only.
set -e -o pipefail
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 50 --warmup 5 --no-cpu --no-extras"
STEADY="python3 $R/bench.py --no-cpu --no-extras"      # the default 2000-step run, for the duration that must agree with bench.py
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $STEADY > $OUT/stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmcA -- $BENCH > $OUT/pmcA.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcB -- $BENCH > $OUT/pmcB.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcC -- $BENCH > $OUT/pmcC.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmcD -- $BENCH > $OUT/pmcD.log 2>&1
cd $R
cat $OUT/stats/*/*_kernel_stats.csv > $OUT/kernel_stats.csv
python3 scripts/pmc_summary.py "$OUT/pmc*/*/*_counter_collection.csv" > $OUT/pmc_summary.txt
cat $OUT/kernel_stats.csv
grep -h "^{" $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json || true
cat $OUT/pmc_summary.txt
python3 scripts/pmc_traffic_record.py $OUT > $OUT/pmc_traffic.json || true
