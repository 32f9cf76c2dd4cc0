# fused frame / frame window at small batches: one-wave kernels (FBUS_TEAM_FRAME=1) against the team kernel (=2)
mkdir -p gpurun_out/r03
python -m pytest tests/test_team_gpu.py -q -x -s -k "frame" 2>&1 | grep -E "team vs|passed|failed|Error|assert" | tail -20
out=gpurun_out/r03/team_frame.txt
: > $out
for B in 256 1024 4096 8192 16384 24576 32768 40960 49152; do
  for T in 1 2; do
    FBUS_TEAM_FRAME=$T python bench.py --batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$B team_frame=$T  per-call %.4g  fused frame %.4g  window %.4g' % (d['value'], d['fused_frame']['value'], d['fused_window']['value']))" >> $out
  done
done
cat $out
