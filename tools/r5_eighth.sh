#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
echo "== stagger for predict_n (fp32, unparked) and frame2_kernel<double>: main = 0 / 0, x2 = 32 / 64, x1 = 64 / 128"
for rep in 1 2; do
for V in main x2 x1; do
  if [ $V = main ]; then unset FBUS_EKF_LIB; else export FBUS_EKF_LIB=$PWD/fbus-ekf_amd/lib/ab/libfbus_$V.so; fi
  timeout 300 python tools/time_predict_n.py 7 65536 2>&1 | grep predict_n | sed "s/^/$V /"
  timeout 300 python tools/run_f64_fused.py --dtype 64 2>&1 | grep -i "frame" | sed "s/^/$V /" | cut -c1-200
done
done 2>&1 | tee $O/stagger_predn_f64.txt
unset FBUS_EKF_LIB
timeout 900 python tools/run_configs.py > $O/run_configs.txt 2>&1; grep -v amdgpu $O/run_configs.txt | cut -c1-200
timeout 1800 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -3 $O/pytest_gpu.log | cut -c1-200; grep "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench.json").read().strip().splitlines()[-1])
print("value %.4g  frac %.3f traffic %s (%s)  hbm frac %.3f traffic %s (%s)  fused_frame %.4g  fused_window %.4g" % (
    d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], d['roofline_hbm_resident']['frac'],
    d['roofline_hbm_resident']['traffic'], d['roofline_hbm_resident']['traffic_source'], d['fused_frame']['value'], d['fused_window']['value']))
print("fp64", d['fp64']['value'], d['fp64']['roofline']['traffic_source'], d['fp64']['roofline_hbm_resident']['traffic_source'])
for k, v in d['north_star_rows'].items():
    if isinstance(v, dict): print(k, '%.3e' % v['value'], v.get('update_avg_launch_us', v.get('frame_avg_launch_us')), v.get('valu_issue_frac'))
print(d['legs_skipped'])
PY
