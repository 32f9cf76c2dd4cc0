cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03/run_configs
rm -rf $OUT; mkdir -p $OUT
export FBUS_RUN_CONFIGS_JSON=$OUT/cases.json
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/run_configs.py > $OUT/hip_events.txt 2> $OUT/err.log
python3 tools/run_configs_rocprof.py $OUT > $OUT/kernel_trace.txt 2>> $OUT/err.log
cat $OUT/hip_events.txt; cat $OUT/kernel_trace.txt; tail -3 $OUT/err.log
