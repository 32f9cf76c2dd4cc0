#!/bin/bash
# A/B of differently built libraries on the reprojection-row update: tools/r4_pix_ab.sh lib1.so lib2.so ...
mkdir -p gpurun_out/r04
for v in "$@"; do
  echo "== $v"
  FBUS_EKF_LIB=$PWD/$v python3 tools/run_pixels.py --both 2>&1 | grep correct_pixels
  FBUS_EKF_LIB=$PWD/$v python3 tools/run_pixels.py --both --batch 16384 --slots 4 2>&1 | grep correct_pixels
done | tee -a gpurun_out/r04/pix_ab.txt
