#!/bin/bash
# round 4 profile sets: headline (fp32, 65 536 filters) and the fp64 legs (65 536 filters; 524 288 filters = 839 MB of 1600-byte records)
mkdir -p gpurun_out/r04
python bench.py --steps 6 --warmup 2 --no-hbm-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['fp64']
print('fp64 per-call %.4g steps/s, predict %.2f us, correct %.2f us; fused frame %.4g steps/s (x%.2f)' % (f['value'], f['roofline']['avg_launch_us'], f['correct_kernel']['avg_launch_us'], f['fused_frame']['value'], f['fused_frame']['vs_per_call']))" | tee gpurun_out/r04/fp64_leg.txt
./tools/profile_gpu.sh r04_b65536 "--batch 65536 --steps 20 --warmup 5" > /dev/null 2>&1
./tools/profile_gpu.sh r04_f64_b65536 "--dtype 64 --batch 65536 --steps 6 --warmup 2" > /dev/null 2>&1
./tools/profile_gpu.sh r04_f64_b524288 "--dtype 64 --batch 524288 --tile 8 --steps 2 --warmup 1" > /dev/null 2>&1
for t in r04_b65536 r04_f64_b65536 r04_f64_b524288; do echo "== $t"; cat gpurun_out/prof_$t/summary.txt | head -40; done
