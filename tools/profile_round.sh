#!/bin/bash
# The profile sets of a round (kernel trace + FETCH_SIZE / WRITE_SIZE PMC passes of bench.py, each in its own rocprofv3 run):
#   rNN_b65536       headline (fp32, 65 536 filters: records cache-resident)
#   rNN_b1048576     fp32 past the Infinity Cache (839 MB of records)
#   rNN_f64_b65536 / rNN_f64_b524288   the fp64 legs
# then the SQ counters of the reprojection-row update (tools/pixels_prof.sh -> gpurun_out/rNN/pix_a/pixels_sq.json).
#   tools/profile_round.sh 06
# afterwards, in the dev container:  for t in r06_b65536 r06_b1048576 r06_f64_b65536 r06_f64_b524288; do python tools/profile_collect.py $t; done
R=${1:-06}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r$R
./tools/profile_gpu.sh r${R}_b65536 "--batch 65536 --steps 20 --warmup 5" > /dev/null 2>&1
./tools/profile_gpu.sh r${R}_b1048576 "--batch 1048576 --tile 16 --steps 3 --warmup 1" > /dev/null 2>&1
./tools/profile_gpu.sh r${R}_f64_b65536 "--dtype 64 --batch 65536 --steps 6 --warmup 2" > /dev/null 2>&1
./tools/profile_gpu.sh r${R}_f64_b524288 "--dtype 64 --batch 524288 --tile 8 --steps 2 --warmup 1" > /dev/null 2>&1
for t in r${R}_b65536 r${R}_b1048576 r${R}_f64_b65536 r${R}_f64_b524288; do echo "== $t"; head -40 gpurun_out/prof_$t/summary.txt; done
bash tools/pixels_prof.sh a $R > gpurun_out/r$R/pixels_prof.log 2>&1; tail -30 gpurun_out/r$R/pixels_prof.log
