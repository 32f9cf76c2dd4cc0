#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two separate PMC passes of the bench.
# Writes raw output under gpurun_out/prof_$TAG and a digest gpurun_out/prof_$TAG/summary.txt.
TAG=${1:-r01}
ARGS=${2:-"--steps 20 --warmup 5 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
python3 tools/profile_digest.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
