#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
tools/_build/exp_lds_dma
python -m pytest tests/test_pixels_gpu.py tests/test_frame_meas_gpu.py tests/test_vision_gpu.py tests/test_policy_gpu.py -m gpu -q -x 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -5
for rep in 1 2; do
for v in fbus-ekf_amd/lib/libfbus_ekf.so fbus-ekf_amd/lib/ab/libfbus_ekf_nopf.so; do
  echo "== $v"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --slots 16 --both --reps 20 2>&1 | grep -a "us per launch"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --slots 4 --both --reps 20 2>&1 | grep -a "us per launch"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --slots 16 --corners --reps 20 2>&1 | grep -a "us per launch"
  FBUS_EKF_LIB=$PWD/$v python tools/run_pixels.py --slots 4 --corners --reps 20 2>&1 | grep -a "us per launch"
done
done
