cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/final; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
wc -l $O/bench.json
for b in 4096 16384 32768 49152 73728 131072; do python3 bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-leg --no-extra-legs > $O/bench_$b.json 2>/dev/null; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03/final/bench*.json')):
    lines = open(f).read().strip().splitlines()
    assert len(lines) == 1, (f, len(lines))
    d = json.loads(lines[0])
    print(f.split('/')[-1], d['config']['batch_per_gpu'], '%.3e' % d['value'], 'predict %.2f us' % d['roofline']['avg_launch_us'], 'correct %.2f us' % d['correct_kernel']['avg_launch_us'], 'fused %.3e window %.3e' % (d['fused_frame']['value'], d['fused_window']['value']), d.get('gather_via'), '%.3f ms' % d['gather_ms'])
PY
python3 -m pytest tests/test_bench_frontdoor.py -m gpu -x -q 2>&1 | tail -3
