#!/bin/bash
# round 5, first GPU call: the new fused frame + north-star update (tests, bench rows), then the whole GPU suite and the default bench
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_frame_meas_gpu.py -x -q -s > $O/frame_meas_tests.log 2>&1; echo "frame_meas tests rc=$?"
tail -5 $O/frame_meas_tests.log
timeout 600 python bench.py --only-pixels > $O/north_star_rows.json 2> $O/north_star_rows.err; echo "only-pixels rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/north_star_rows.json").read().strip().splitlines()[-1])["north_star_rows"]
for k, v in d.items():
    if isinstance(v, dict):
        print(f"{k:32s} {v['value']:.4g} steps/s  {v.get('update_avg_launch_us', v.get('frame_avg_launch_us')):.1f} us  applied {v['filters_updated_frac']:.3f} finite {v['state_finite']}")
PY
timeout 1500 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -4 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r05/bench.json').read().strip().splitlines()[-1])
print('value %.4g  ms/step %.3f  frac %.3f  fused_frame %.4g  fused_window %.4g' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['fused_frame']['value'], d['fused_window']['value']))"
