bash tools/r3_run_configs.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/final; mkdir -p $O
python3 tools/run_configs.py > $O/run_configs_hip_events.txt 2>/dev/null
cat $O/run_configs_hip_events.txt; cat gpurun_out/r03/run_configs/kernel_trace.txt
