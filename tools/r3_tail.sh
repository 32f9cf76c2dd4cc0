# tail rounds: batches between 1024 and 2048 waves with the one-wave-per-SIMD instantiations (default) against the two-wave ones
mkdir -p gpurun_out/r03
out=gpurun_out/r03/tail_sweep.txt
: > $out
for B in 69632 73728 81920 98304 114688; do
  for T in default 65600; do
    if [ $T = default ]; then unset FBUS_TWO_WAVE_MIN_B; else export FBUS_TWO_WAVE_MIN_B=$T; fi
    python bench.py --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$B two_wave_min=$T  %.4g  predict %.2f  correct %.2f  fused %.4g  window %.4g' % (d['value'], d['roofline']['avg_launch_us'], d['correct_kernel']['avg_launch_us'], d['fused_frame']['value'], d['fused_window']['value']))" >> $out
  done
done
cat $out
