mkdir -p gpurun_out/r03
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_a.json 2> gpurun_out/r03/bench_a.err ) 2> gpurun_out/r03/bench_a.time
tail -3 gpurun_out/r03/bench_a.time
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/bench_a.json').read().strip().splitlines()[-1])
print(d['metric'], '%.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'])
r = d['roofline']; print('predict us %.2f frac %.3f' % (r['avg_launch_us'], r['frac']), 'correct us %.2f frac %.3f' % (d['correct_kernel']['avg_launch_us'], d['correct_kernel']['frac']))
print('fused %.4g window %.4g' % (d['fused_frame']['value'], d['fused_window']['value']))
for k in ('roofline_hbm_resident', 'roofline_b262144'):
    h = d[k]
    if h: print(k, 'value %.4g predict us %.2f achieved %.0f frac %.3f of copy %.3f correct us %.2f frac %.3f fused %.4g' % (h['value'], h['avg_launch_us'], h['achieved'], h['frac'], h['frac_of_copy_ceiling'], h['correct_kernel']['avg_launch_us'], h['correct_kernel']['frac'], h['fused_frame_value']))
f = d['fp64']
if f: print('fp64 value %.4g predict us %.2f frac %.3f correct us %.2f frac %.3f' % (f['value'], f['roofline']['avg_launch_us'], f['roofline']['frac'], f['correct_kernel']['avg_launch_us'], f['correct_kernel']['frac']))
print('compute_bound', d['compute_bound'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
tail -5 gpurun_out/r03/bench_a.err
