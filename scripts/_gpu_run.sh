set -o pipefail
mkdir -p gpurun_out/r4g
timeout -k 10 700 python scripts/walk_soak.py --seconds 420 2>&1 | tee gpurun_out/r4g/walk_soak.txt | tail -6
