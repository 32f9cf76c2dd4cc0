#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two separate PMC passes of the bench (the program itself
# after `--`, never a wrapper).  Writes raw output under gpurun_out/prof_$TAG and a digest
# gpurun_out/prof_$TAG/summary.txt + digest.json (copy those into profiles/).
#   tools/profile_gpu.sh r02_b65536  "--batch 65536 --steps 20 --warmup 5"
#   tools/profile_gpu.sh r02_b262144 "--batch 262144 --steps 4 --warmup 1"
#   tools/profile_gpu.sh r03_b1048576 "--batch 1048576 --tile 16 --steps 3 --warmup 1"     (839 MB of records: past the Infinity Cache)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the GPU box exports GRAFT_REPO_ROOT; elsewhere: the script's repository
TAG=${1:-r02_b65536}
ARGS=${2:-"--batch 65536 --steps 20 --warmup 5"}
ARGS="$ARGS --no-cpu-baseline --no-hbm-leg --no-extra-legs"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout 120 python3 tools/profile_digest.py $OUT "$ARGS" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
