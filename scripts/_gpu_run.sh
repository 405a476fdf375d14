set -o pipefail
mkdir -p gpurun_out/r3e
for v in full norot full norot full norot; do
  if [ "$v" = full ]; then lib=evidence_amd/librvll.so; else lib=evidence_amd/diag/librvll_$v.so; fi
  RVLL_LIBRARY=$PWD/$lib python scripts/long_solve_tail.py $v 2>&1 | tee -a gpurun_out/r3e/tail.txt
  RVLL_LIBRARY=$PWD/$lib python bench.py --no-cpu --no-extras > gpurun_out/r3e/bench_$v.json 2>/dev/null
  python scripts/show_bench_keys.py gpurun_out/r3e/bench_$v.json | cut -c1-150 | tee -a gpurun_out/r3e/tail.txt
done
python -m pytest tests/test_gpu_loglike.py tests/test_gpu_forms.py tests/test_gpu_precision.py -m gpu -x -q 2>&1 | tail -3
