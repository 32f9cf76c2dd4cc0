# the complete round-3 evidence set in one GPU call (final build): see profiles/README.md
bash tools/profile_gpu.sh r03_b65536 "--batch 65536 --steps 20 --warmup 5" > /dev/null 2>&1
bash tools/profile_gpu.sh r03_b1048576 "--batch 1048576 --tile 16 --steps 3 --warmup 1" > /dev/null 2>&1
bash tools/pmc_pixels.sh > /dev/null 2>&1
bash tools/r3_run_configs.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/final; rm -rf $O; mkdir -p $O
python3 tools/run_configs.py > $O/run_configs_hip_events.txt 2>/dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/team -- python3 tools/time_team.py 4096 16384 32768 > $O/team_hip_events.txt 2>/dev/null
python3 tools/trace_team.py $O/team > $O/team_kernel_trace.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
for b in 4096 16384 32768 49152 73728 131072 262144; do python3 bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-leg --no-extra-legs > $O/bench_$b.json 2>/dev/null; done
bash tools/pmc_sq.sh r03_b65536 > /dev/null 2>&1
grep -E "predict_kernel|correct_kernel|frame" gpurun_out/prof_r03_b65536/summary.txt | head -12
grep -E "predict_kernel|correct_kernel|frame" gpurun_out/prof_r03_b1048576/summary.txt | head -12
cat gpurun_out/r03/run_configs/kernel_trace.txt
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03/final/bench*.json')):
    lines = open(f).read().strip().splitlines()
    d = json.loads(lines[0])
    print(f.split('/')[-1], d['config']['batch_per_gpu'], '%.3e' % d['value'], 'predict %.2f us' % d['roofline']['avg_launch_us'], 'correct %.2f us' % d['correct_kernel']['avg_launch_us'], 'fused %.3e window %.3e' % (d['fused_frame']['value'], d['fused_window']['value']))
PY
