#!/bin/bash
# fp32 measurement kernels (per call, divided tail, fused frame) built with another scheduling strategy: tools/ab_meas32.sh libA.so libB.so
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in 1 2; do for v in "$@"; do
  echo "== $v"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --both 2>&1 | grep "correct_"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --corners 2>&1 | grep "correct_"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --batch 16384 --slots 4 --both 2>&1 | grep "correct_"
  FBUS_EKF_LIB=$PWD/$v python bench.py --only-pixels 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); n=d.get('north_star_rows', d)
print('  '.join('%s %.4g' % (k, n[k]['value']) for k in ('pixels_m4','pixels_m4_stereo','fused_frame_pixels_m4','fused_frame_pixels_m4_stereo','fused_frame_corners_m4','fused_window_pixels_m4') if k in n))"
done; done 2>&1 | tee gpurun_out/r05/ab_meas32.txt
