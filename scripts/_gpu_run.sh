set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python scripts/scalar_latency_ab.py full
python scripts/scalar_latency_ab.py full
python bench.py --no-cpu --no-extras | python scripts/show_bench_keys.py /dev/stdin | cut -c1-140
