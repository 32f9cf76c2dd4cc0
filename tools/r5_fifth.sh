#!/bin/bash
# round 5, fourth GPU call: per-SIMD stagger A/B (library variants under fbus-ekf_amd/lib/ab), fused-frame tests, split-kernel A/B repeated
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
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
for rep in 1; do
for V in main stgD stgC stgA stgB; do
  if [ $V = main ]; then unset FBUS_EKF_LIB; else export FBUS_EKF_LIB=$PWD/fbus-ekf_amd/lib/ab/libfbus_$V.so; fi
  echo "== $V (rep $rep)"
  timeout 600 python bench.py --only-pixels --no-hbm-leg > $O/ns_$V.json 2> $O/ns_$V.err && rows $O/ns_$V.json
  timeout 600 python bench.py --no-hbm-leg --no-cpu-baseline --no-extra-legs > $O/b_$V.json 2> $O/b_$V.err && main $O/b_$V.json
done
done 2>&1 | tee $O/stagger_sweep.txt
unset FBUS_EKF_LIB
