#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_frame_meas_gpu.py -q -s > $O/frame_meas_tests.log 2>&1; echo "frame_meas tests rc=$?"
grep "parity\|passed\|failed\|Error\|assert " $O/frame_meas_tests.log | cut -c1-260 | head -60
timeout 600 python bench.py --only-pixels > $O/north_star_rows.json 2> $O/north_star_rows.err; echo "only-pixels rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/north_star_rows.json").read().strip().splitlines()[-1])["north_star_rows"]
for k, v in d.items():
    if isinstance(v, dict):
        print(f"{k:32s} {v['value']:.4g} steps/s  {v.get('update_avg_launch_us', v.get('frame_avg_launch_us')):.1f} us  applied {v['filters_updated_frac']:.3f} finite {v['state_finite']}")
PY
echo "== tail timeline"
TAIL_B=69632 timeout 120 ./tools/_build/exp_timeline 2>&1 | tee $O/tail_timeline.txt
TAIL_B=65600 timeout 120 ./tools/_build/exp_timeline 2>&1 | tee -a $O/tail_timeline.txt
echo "== split times"
for S in 0 2 4; do
  export FBUS_MEAS_SPLIT=$S
  echo "-- FBUS_MEAS_SPLIT=$S"
  timeout 300 python tools/run_pixels.py --both --batch 16384 --slots 4 --roles $( [ $S = 0 ] && echo 0 || echo 0 ) 2>&1 | grep correct_
  timeout 300 python tools/run_pixels.py --both --batch 32768 --slots 4 2>&1 | grep correct_
  timeout 300 python tools/run_pixels.py --both --batch 32768 --slots 16 2>&1 | grep correct_
  timeout 300 python tools/run_pixels.py --both --batch 65536 --slots 4 2>&1 | grep correct_
done 2>&1 | tee $O/split_times2.txt
unset FBUS_MEAS_SPLIT
timeout 1800 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -4 $O/pytest_gpu.log; grep "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
