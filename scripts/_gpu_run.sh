set -o pipefail
R=$PWD; OUT=$R/gpurun_out/r3m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rows in 1 0; do
  export RVLL_WALK_ROWS=$rows
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$rows -- python3 $R/scripts/walk_once.py 0.9 > $OUT/stats$rows.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc$rows -- python3 $R/scripts/walk_once.py 0.9 > $OUT/pmc$rows.log 2>&1
  echo "== rows=$rows"; tail -2 $OUT/stats$rows.log
  cat $OUT/stats$rows/*/*_kernel_stats.csv | grep -E "slice_walk|Name" | cut -c1-200
  python3 $R/scripts/pmc_summary.py "$OUT/pmc$rows/*/*_counter_collection.csv" slice_walk
done
