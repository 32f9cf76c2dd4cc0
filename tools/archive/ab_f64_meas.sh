#!/bin/bash
# fp64 measurement kernels built with other scheduling strategies (fewer bytes of scratch): tools/ab_f64_meas.sh libA.so libB.so ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for rep in 1 2; do for v in "$@"; do
  echo "== $v"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --dtype 64 --both 2>&1 | grep "correct_"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --dtype 64 --corners 2>&1 | grep "correct_"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --dtype 64 --batch 16384 --slots 4 2>&1 | grep "correct_"
done; done 2>&1 | tee gpurun_out/r05/ab_f64_meas.txt
