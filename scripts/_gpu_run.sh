set -o pipefail
mkdir -p gpurun_out/r3s
timeout -k 10 900 python -m pytest tests/test_gpu_loglike.py tests/test_gpu_forms.py tests/test_gpu_precision.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -3
for v in full nologdet full nologdet; do
  if [ "$v" = full ]; then lib=evidence_amd/librvll.so; else lib=evidence_amd/diag/librvll_$v.so; fi
  echo "# $v" | tee -a gpurun_out/r3s/sweep.txt
  RVLL_LIBRARY=$PWD/$lib python scripts/form_sweep.py 2>&1 | grep -E "^  (3     200   16384|4    1000    8192|5    2000   16384|2     200    4096|5    2000     256)" | tee -a gpurun_out/r3s/sweep.txt
done
