import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l); print('%.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'predict %.2f us' % d['roofline']['avg_launch_us'], 'correct %.2f' % d['correct_kernel']['avg_launch_us'], 'fused %.4g' % d['fused_frame']['value'], 'window %.4g' % (d.get('fused_window') or {}).get('value', float('nan')))
