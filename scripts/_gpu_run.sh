set -o pipefail
mkdir -p gpurun_out/r4d
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4d/gputests.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 gpurun_out/r4d/gputests.log | cut -c1-200
python3 bench.py > gpurun_out/r4d/bench_default.json 2> gpurun_out/r4d/bench_default.err; echo "bench default rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4d/bench_driver_args.json 2> gpurun_out/r4d/bench_driver_args.err; echo "bench driver rc=$?"
python3 scripts/show_bench_keys.py gpurun_out/r4d/bench_default.json gpurun_out/r4d/bench_driver_args.json | grep -E "value|nested" | cut -c1-420
