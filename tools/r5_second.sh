#!/bin/bash
# round 5, second GPU call: diagnostics of the fused frame and of the long-run finiteness; the divided-update pixel kernel (parity, times);
# the sub-tile tail of the per-call predict
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
timeout 600 python tools/r5_diag.py ab > $O/diag.txt 2>&1; echo "diag rc=$?"; cat $O/diag.txt | grep -v amdgpu.ids
echo "== pixel tests (split kernel behind correct_roles 0 / 2)"
timeout 900 python -m pytest tests/test_pixels_gpu.py -x -q -s > $O/pixels_tests.log 2>&1; echo "pixels tests rc=$?"; grep "parity\|passed\|failed\|Error" $O/pixels_tests.log | cut -c1-230
echo "== times: split (default) vs FBUS_MEAS_SPLIT=0"
for S in 0 auto; do
  if [ $S = auto ]; then unset FBUS_MEAS_SPLIT; else export FBUS_MEAS_SPLIT=$S; fi
  echo "-- FBUS_MEAS_SPLIT=$S"
  timeout 300 python tools/run_pixels.py --both 2>&1 | grep correct_
  timeout 300 python tools/run_pixels.py --both --slots 4 2>&1 | grep correct_
  timeout 300 python tools/run_pixels.py --both --batch 16384 --slots 4 2>&1 | grep correct_
  timeout 300 python tools/run_pixels.py --both --batch 32768 --slots 4 2>&1 | grep correct_
done 2>&1 | tee $O/split_times.txt
unset FBUS_MEAS_SPLIT
echo "== predict tail"
for B in 65536 65600 66560 69632 73728 81920 98304 131136; do
  for S in 0 1; do TAIL_POL=1 TAIL_B=$B TAIL_SPLIT=$S timeout 120 ./tools/_build/exp_timeline 2>&1 | grep "nt loads, nt stores\|default loads, default"; done
done | tee $O/tail_split.txt
