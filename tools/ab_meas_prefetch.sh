#!/bin/bash
# A/B of library builds on the reprojection-row / corner-row updates (HIP-event bracket per launch), alternating:
#   tools/ab_meas_prefetch.sh libA.so libB.so ...        (paths relative to the repository; default: the in-tree library and lib/ab/*.so)
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
LIBS=${@:-fbus-ekf_amd/lib/libfbus_ekf.so $(ls fbus-ekf_amd/lib/ab/*.so 2>/dev/null)}
[ -x tools/_build/exp_lds_dma ] && tools/_build/exp_lds_dma
for rep in 1 2 3; do
for v in $LIBS; do
  echo "== $v"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --slots 16 --both --reps 20 2>&1 | grep -a "us per launch"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --slots 4 --both --reps 20 2>&1 | grep -a "us per launch"
done
done
