#!/bin/bash
# the resident ImuUpdate loop at 65 536 filters: one wave per tile (roles 1) against two (predict_n_duo_kernel, roles 2), library variants
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in 1 2; do
  TEAM_PREDICT=1 python tools/time_predict_n.py 18 2>&1 | grep step
  for v in "$@"; do FBUS_EKF_LIB=$PWD/$v TEAM_PREDICT=2 python tools/time_predict_n.py 18 2>&1 | grep step; done
done 2>&1 | tee gpurun_out/r05/ab_duo.txt
