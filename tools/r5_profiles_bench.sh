#!/bin/bash
# the four bench.py profile sets of tools/r5_profiles.sh without the pixel-kernel passes (re-collected on the final tree of the round)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
./tools/profile_gpu.sh r05_b65536 "--batch 65536 --steps 20 --warmup 5" > /dev/null 2>&1
./tools/profile_gpu.sh r05_b1048576 "--batch 1048576 --tile 16 --steps 3 --warmup 1" > /dev/null 2>&1
./tools/profile_gpu.sh r05_f64_b65536 "--dtype 64 --batch 65536 --steps 6 --warmup 2" > /dev/null 2>&1
./tools/profile_gpu.sh r05_f64_b524288 "--dtype 64 --batch 524288 --tile 8 --steps 2 --warmup 1" > /dev/null 2>&1
for t in r05_b65536 r05_b1048576 r05_f64_b65536 r05_f64_b524288; do echo "== $t"; head -12 gpurun_out/prof_$t/summary.txt; done
