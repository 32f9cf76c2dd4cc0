mkdir -p gpurun_out/r03
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_q.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/bench_q.json').read().strip().splitlines()[-1])
print('%.4g' % d['value'], 'predict us %.2f' % d['roofline']['avg_launch_us'], 'correct us %.2f' % d['correct_kernel']['avg_launch_us'], 'fused %.4g window %.4g' % (d['fused_frame']['value'], d['fused_window']['value']))
h = d['roofline_hbm_resident']; print('1M: %.4g predict %.1f correct %.1f fused %.4g' % (h['value'], h['avg_launch_us'], h['correct_kernel']['avg_launch_us'], h['fused_frame_value']))
PY
