set -o pipefail
mkdir -p gpurun_out/r3z
bash scripts/profile_gpu.sh r03prof2 > gpurun_out/r3z/profile_gpu.log 2>&1; echo "profile_gpu rc=$?"; tail -22 gpurun_out/r3z/profile_gpu.log | cut -c1-200
python3 bench.py > gpurun_out/r3z/bench_default.json 2> gpurun_out/r3z/bench_default.err; echo "bench default rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3z/bench_driver_args.json 2> gpurun_out/r3z/bench_driver_args.err; echo "bench driver rc=$?"
python3 scripts/show_bench_keys.py gpurun_out/r3z/bench_default.json gpurun_out/r3z/bench_driver_args.json | grep -E "value" | cut -c1-260
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
