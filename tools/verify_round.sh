#!/bin/bash
# the driver's round-end sequence on the current tree: GPU suite, smoke, default bench (compact line + bench_detail.json)
#   tools/verify_round.sh 06        -> gpurun_out/r06/{pytest_gpu.log, bench_line.json, bench_detail.json, bench.err}
R=${1:-06}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r$R
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
grep -a "passed\|failed" $O/pytest_gpu.log | tail -2 | cut -c1-200; grep -a "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 | cut -c1-200
timeout 900 python bench.py --detail-file $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l = open('$O/bench_line.json').read().strip().splitlines()[-1]
d = json.loads(l)
print('line bytes', len(l), ' value %.4g  frac %.3f  hbm frac %.3f  cpu %.3g  fused pixels %.4g' % (
    d['value'], d['roofline']['frac'], d['roofline']['hbm_resident']['frac'], d['cpu_baseline']['value'], d['north_star']['fused_frame_pixels_m4']['value']))
PY
