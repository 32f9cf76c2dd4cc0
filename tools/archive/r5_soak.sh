#!/bin/bash
# end-of-round soak: the GPU suite three times (flakiness), the default bench five times (spread of the headline and the fused legs)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r05
for i in 1 2 3; do timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1; done
for i in 1 2 3 4 5; do
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.4g  ms_per_step %.4f  roofline.frac %.3f  hbm %.3f  fused_frame %.4g  fused_window %.4g  fused pixels m4 %.4g' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm_resident']['frac'], d['fused_frame']['value'], d['fused_window']['value'], d['north_star_rows']['fused_frame_pixels_m4']['value']))"
done 2>&1 | tee gpurun_out/r05/soak.txt
