# large batches (two-wave kernel forms): row-split correct, frame2, window
python -m pytest tests/test_parity_gpu.py -q -x -k "row_split or two_wave" 2>&1 | tail -2
for B in 73728 131072 262144; do python bench.py --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['batch_per_gpu'], '%.4g' % d['value'], 'correct %.2f' % d['correct_kernel']['avg_launch_us'], 'fused %.4g window %.4g' % (d['fused_frame']['value'], d['fused_window']['value']))"; done
