#!/bin/bash
# the driver's round-end sequence on the final tree: GPU suite, smoke, default bench
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -3 $O/pytest_gpu.log | cut -c1-200; grep "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 | cut -c1-200
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/r05/bench.json').read().strip().splitlines()[-1])
print('value %.4g  frac %.3f  hbm frac %.3f  fused_frame %.4g  fused_window %.4g  fused pixels %.4g' % (d['value'], d['roofline']['frac'], d['roofline_hbm_resident']['frac'], d['fused_frame']['value'], d['fused_window']['value'], d['north_star_rows']['fused_frame_pixels_m4']['value']))"
