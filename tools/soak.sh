#!/bin/bash
# end-of-round soak: the GPU suite three times (flakiness), the default bench five times (spread of the headline and the fused legs;
# every run's stdout line must parse and stay below 6 KB)      tools/soak.sh 06 -> gpurun_out/r06/soak.txt
R=${1:-06}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r$R
{
for i in 1 2 3; do timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -aE "passed|failed" | tail -1; done
for i in 1 2 3 4 5; do
  python bench.py --no-cpu-baseline --detail-file /tmp/soak_detail.json 2>/dev/null | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip(); d=json.loads(l)
print('line %d B  value %.4g  ms_per_step %.4f  roofline.frac %.3f  hbm %.3f  fused_frame %.4g  fused_window %.4g  fused pixels m4 %.4g' % (len(l), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['hbm_resident']['frac'], d['fused_frame']['value'], d['fused_window']['value'], d['north_star']['fused_frame_pixels_m4']['value']))"
done
} 2>&1 | tee gpurun_out/r$R/soak.txt
