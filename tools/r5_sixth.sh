#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
O=gpurun_out/r05
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"
tail -3 $O/pytest_gpu.log | cut -c1-200; grep "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep parity $O/smoke.log | cut -c1-200
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
for V in main sc1; do
  if [ $V = main ]; then unset FBUS_EKF_LIB; else export FBUS_EKF_LIB=$PWD/fbus-ekf_amd/lib/ab/libfbus_$V.so; fi
  echo "== $V (rep $rep)"
  timeout 600 python bench.py --only-pixels --no-hbm-leg > $O/ns_$V.json 2> $O/ns_$V.err && rows $O/ns_$V.json
  timeout 600 python bench.py --no-hbm-leg --no-cpu-baseline --no-extra-legs > $O/b_$V.json 2> $O/b_$V.err && main $O/b_$V.json
done
done 2>&1 | tee $O/frame_st_ab.txt
unset FBUS_EKF_LIB
out=$O/bench_by_batch.txt
echo "bench.py --batch B --steps 6 --warmup 2 (65 536: --steps 20 --warmup 5), one MI355X, un-profiled; launch times: HIP events around a step of the batch" > $out
echo " filters   EKF steps/s  predict us  correct us  fused frame  frame window" >> $out
for B in 65536 69632 73728 81920 98304 131072 262144; do
  S="--steps 6 --warmup 2"; [ $B = 65536 ] && S="--steps 20 --warmup 5"
  python bench.py --batch $B $S --no-cpu-baseline --no-extra-legs --no-hbm-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%8d    %.3e  %10.2f  %10.2f    %.3e    %.3e' % ($B, d['value'], d['roofline']['avg_launch_us'], d['correct_kernel']['avg_launch_us'], d['fused_frame']['value'], d['fused_window']['value']))" >> $out
done
cat $out
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; main $O/bench.json
