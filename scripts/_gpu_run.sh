set -o pipefail
mkdir -p gpurun_out/r3n
timeout -k 10 600 python -m pytest tests/test_gpu_walk.py -m gpu -x -q > gpurun_out/r3n/walktests.log 2>&1; rc=$?; echo "walk tests rc=$rc"; tail -12 gpurun_out/r3n/walktests.log
[ $rc -eq 0 ] || exit 1
python bench.py --no-cpu > gpurun_out/r3n/bench.json 2> gpurun_out/r3n/bench.err; echo "bench rc=$?"
python scripts/show_bench_keys.py gpurun_out/r3n/bench.json
