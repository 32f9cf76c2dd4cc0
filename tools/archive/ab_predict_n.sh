#!/bin/bash
# A/B of library builds on the resident ImuUpdate loop: tools/ab_predict_n.sh libA.so libB.so ...
# (FBUS_TWO_WAVE_MIN_B=1 forces the <= 256-register forms -- predict_n with rows p and the nominal state parked in LDS -- at every batch)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in 1 2; do for v in "$@"; do for tw in "" 1; do
  echo "FBUS_TWO_WAVE_MIN_B=$tw"
  FBUS_TWO_WAVE_MIN_B=$tw FBUS_EKF_LIB=$PWD/$v python tools/time_predict_n.py 18 2>&1 | grep step
  FBUS_TWO_WAVE_MIN_B=$tw FBUS_EKF_LIB=$PWD/$v python tools/time_predict_n.py 18 --dialect 1 2>&1 | grep step
done; done; done 2>&1 | tee gpurun_out/r05/ab_predict_n.txt
