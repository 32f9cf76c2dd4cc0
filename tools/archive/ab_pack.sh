#!/bin/bash
# A/B of library builds inside ONE gpurun call, the VALU-bound legs: tools/ab_pack.sh libA.so libB.so ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  FBUS_EKF_LIB=$PWD/$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
n=d.get('north_star_rows',{})
print('$v'.split('/')[-1].ljust(18), 'value %.4g' % d['value'], ' fused_frame %.4g' % d['fused_frame']['value'], ' fused_window %.4g' % d['fused_window']['value'],
      ' fused pixels m4 %.4g' % n.get('fused_frame_pixels_m4',{}).get('value',0), ' window pixels %.4g' % n.get('fused_window_pixels_m4',{}).get('value',0))"
done; done 2>&1 | tee gpurun_out/r05/ab_pack.txt
