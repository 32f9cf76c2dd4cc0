#!/bin/bash
# the bench legs by batch size across the step behind 65 536 filters      tools/bench_by_batch.sh 06 -> gpurun_out/r06/bench_by_batch.txt
R=${1:-06}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out/r$R
out=gpurun_out/r$R/bench_by_batch.txt
echo "bench.py --batch B --steps 6 --warmup 2 (65 536: --steps 20 --warmup 5), one MI355X, un-profiled; launch times: HIP events around a step of the batch" > $out
echo " filters   EKF steps/s  predict us  correct us  fused frame  frame window" >> $out
for B in ${BATCHES:-32768 65536 69632 73728 81920 98304 131072 262144}; do
  S="--steps 6 --warmup 2"; [ $B = 65536 ] && S="--steps 20 --warmup 5"
  python bench.py --batch $B $S --no-cpu-baseline --no-extra-legs --no-hbm-leg --detail-file /tmp/bb_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%8d    %.3e  %10.2f  %10.2f    %.3e    %.3e' % ($B, d['value'], d['roofline']['avg_launch_us'], d['correct_kernel']['avg_launch_us'], d['fused_frame']['value'], d['fused_window']['value']))" >> $out
done
cat $out
