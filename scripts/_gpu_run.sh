set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for v in full nologdet full nologdet; do
  if [ "$v" = full ]; then lib=evidence_amd/librvll.so; else lib=evidence_amd/diag/librvll_$v.so; fi
  RVLL_LIBRARY=$PWD/$lib python scripts/scalar_latency_ab.py $v
done
python bench.py --no-cpu --no-extras | python scripts/show_bench_keys.py /dev/stdin | cut -c1-140
