#!/bin/bash
# kernel-trace of the bench for several builds of the library in ONE gpurun call, reduced by trace_positions.py
# usage: tools/ab_trace.sh libA.so libB.so ...
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the GPU box exports GRAFT_REPO_ROOT; elsewhere: the script's repository
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
for v in "$@"; do
  OUT=gpurun_out/abtrace/$(basename $v .so); rm -rf $OUT; mkdir -p $OUT
  export FBUS_EKF_LIB=$PWD/$v
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-timing off > $OUT/log 2>&1
  echo "== $v: $(python3 tools/bench_line.py < $OUT/log)"
  python3 tools/trace_positions.py $OUT
done
