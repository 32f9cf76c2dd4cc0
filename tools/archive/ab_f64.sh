#!/bin/bash
# the fp64 legs of bench.py for several library builds: tools/ab_f64.sh libA.so libB.so ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in 1 2 3; do for v in "$@"; do
  FBUS_EKF_LIB=$PWD/$v python bench.py --dtype 64 --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v'.split('/')[-1].ljust(20), 'fp64 value %.4g  predict %.2f us  correct %.2f us  fused_frame %.4g  fused_window %.4g' % (d['value'], d['roofline']['avg_launch_us'], d['correct_kernel']['avg_launch_us'], d['fused_frame']['value'], d['fused_window']['value']))"
done; done 2>&1 | tee gpurun_out/r05/ab_f64.txt
