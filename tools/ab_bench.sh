#!/bin/bash
# A/B of several builds of libfbus_ekf.so inside ONE gpurun call (boxes differ by a few %): alternates A B C A B C.
# usage: tools/ab_bench.sh libA.so libB.so [libC.so ...] [-- bench args]
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
ARGS=${@:-"--steps 60 --warmup 5 --no-cpu-baseline --no-hbm-leg"}
for rep in $(seq 1 ${REPS:-3}); do for v in "${LIBS[@]}"; do
  FBUS_EKF_LIB=$PWD/$v python bench.py $ARGS 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v'.split('/')[-1].ljust(28), '%.4g' % d['value'], 'predict %.2f us' % d['roofline']['avg_launch_us'], 'correct %.2f' % d['correct_kernel']['avg_launch_us'], 'fused %.4g' % d['fused_frame']['value'])"
done; done
