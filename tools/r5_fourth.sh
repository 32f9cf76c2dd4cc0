#!/bin/bash
# round 5, fourth GPU call: per-SIMD stagger A/B (library variants under fbus-ekf_amd/lib/ab), fused-frame tests, split-kernel A/B repeated
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
timeout 900 python -m pytest tests/test_frame_meas_gpu.py -q -s > $O/frame_meas_tests.log 2>&1; echo "frame_meas tests rc=$?"
grep "passed\|failed\|Error" $O/frame_meas_tests.log | cut -c1-200 | head
rows() { python - "$1" <<'PY'
import json, sys
dd = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
d = dd.get("north_star_rows") or {}
print("   " + "  ".join(f"{k}={v['value']:.3e}/{v.get('update_avg_launch_us', v.get('frame_avg_launch_us')):.1f}us" for k, v in d.items() if isinstance(v, dict)))
PY
}
main() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   value %.4g  predict %.2f us  correct %.2f us  fused_frame %.4g  fused_window %.4g" % (d['value'], d['roofline']['avg_launch_us'], d['correct_kernel']['avg_launch_us'], d['fused_frame']['value'], d['fused_window']['value']))
PY
}
for rep in 1 2; do
for V in main stg32 stg64; do
  if [ $V = main ]; then unset FBUS_EKF_LIB; else export FBUS_EKF_LIB=$PWD/fbus-ekf_amd/lib/ab/libfbus_$V.so; fi
  echo "== $V (rep $rep)"
  timeout 600 python bench.py --only-pixels --no-hbm-leg > $O/ns_$V.json 2> $O/ns_$V.err && rows $O/ns_$V.json
  timeout 600 python bench.py --no-hbm-leg --no-cpu-baseline --no-extra-legs > $O/b_$V.json 2> $O/b_$V.err && main $O/b_$V.json
done
done 2>&1 | tee $O/stagger_ab.txt
unset FBUS_EKF_LIB
echo "== split A/B, alternating"
for rep in 1 2 3; do
  for S in 0 2; do FBUS_MEAS_SPLIT=$S timeout 300 python tools/run_pixels.py --batch 32768 --slots 4 2>&1 | grep correct_ | sed "s/^/split=$S /"; done
  for S in 0 4; do FBUS_MEAS_SPLIT=$S timeout 300 python tools/run_pixels.py --batch 16384 --slots 4 2>&1 | grep correct_ | sed "s/^/split=$S /"; done
done 2>&1 | cut -c1-120 | tee $O/split_ab.txt
