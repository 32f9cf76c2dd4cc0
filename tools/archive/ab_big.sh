#!/bin/bash
# A/B of library builds past one round of waves (two waves per SIMD: the 256-register forms): tools/ab_big.sh libA.so libB.so
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in 1 2; do for v in "$@"; do
  FBUS_EKF_LIB=$PWD/$v python tools/time_predict_n.py 18 --batch 262144 2>&1 | grep step
  FBUS_EKF_LIB=$PWD/$v python bench.py --batch 262144 --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v'.split('/')[-1].ljust(18), 'B=262144 value %.4g' % d['value'], ' fused_frame %.4g' % d['fused_frame']['value'], ' fused_window %.4g' % d['fused_window']['value'])"
done; done 2>&1 | tee gpurun_out/r05/ab_big.txt
