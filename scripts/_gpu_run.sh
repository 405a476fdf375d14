set -o pipefail
mkdir -p gpurun_out/r3x
bash scripts/profile_all.sh r03all > gpurun_out/r3x/profile_all.log 2>&1; echo "profile_all rc=$?"; tail -5 gpurun_out/r03all/stats.log | cut -c1-200
python3 bench.py > gpurun_out/r3x/bench_default.json 2> gpurun_out/r3x/bench_default.err; echo "bench default rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3x/bench_driver_args.json 2> gpurun_out/r3x/bench_driver_args.err; echo "bench driver rc=$?"
timeout -k 10 400 python3 bench.py --gpus 2 --steps 200 --warmup 20 > gpurun_out/r3x/n2.json 2> gpurun_out/r3x/n2.err; echo "n2 rc=$?"
python3 scripts/high_ecc_parity.py > gpurun_out/r3x/high_ecc_parity.txt 2>&1; tail -12 gpurun_out/r3x/high_ecc_parity.txt
python3 scripts/show_bench_keys.py gpurun_out/r3x/bench_default.json gpurun_out/r3x/bench_driver_args.json | grep -E "value|nested|host_round" | cut -c1-400
