#!/bin/bash
# round 4: headline by batch, one launch per EKF step (FBUS_CHUNK=0) against whole rounds as launches of their own (default)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/bench_by_batch.txt
echo "bench.py --batch B --steps 6 --warmup 2 (65 536: --steps 20 --warmup 5), one MI355X, un-profiled; launch times: HIP events around a step of the batch" > $out
echo " filters  chunk   EKF steps/s  predict us  correct us  fused frame  frame window" >> $out
for B in ${BATCHES:-65536 69632 73728 98304 131072 196608 262144}; do
  for C in 0 auto; do
    if [ $C = auto ]; then unset FBUS_CHUNK; else export FBUS_CHUNK=$C; fi
    S="--steps 6 --warmup 2"; [ $B = 65536 ] && S="--steps 20 --warmup 5"
    python bench.py --batch $B $S --no-cpu-baseline --no-extra-legs --no-hbm-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%8d %6s    %.3e  %10.2f  %10.2f    %.3e    %.3e' % ($B, d['roofline']['launch_policy']['chunk_filters'] or '-', d['value'], d['roofline']['avg_launch_us'], d['correct_kernel']['avg_launch_us'], d['fused_frame']['value'], d['fused_window']['value']))" >> $out
  done
done
cat $out
