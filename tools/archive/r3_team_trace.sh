cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/team2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/team -- python3 tools/time_team.py 4096 8192 16384 32768 > $O/team_hip_events.txt 2>/dev/null
python3 tools/trace_team.py $O/team > $O/team_kernel_trace.txt
grep -E "predict" $O/team_kernel_trace.txt
