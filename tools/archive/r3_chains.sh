# launch chains x predict cache policy at large batches (tools/exp_two_streams.py; each S printed twice)
mkdir -p gpurun_out/r03
out=gpurun_out/r03/chains.txt
: > $out
for B in 524288 1048576; do
  for P in auto 0 1 2; do
    if [ $P = auto ]; then unset FBUS_PREDICT_POLICY; else export FBUS_PREDICT_POLICY=$P; fi
    echo "policy $P" >> $out
    B=$B python tools/exp_two_streams.py 2>&1 | grep streams | tail -3 >> $out
  done
done
cat $out
