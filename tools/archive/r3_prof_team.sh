cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03/prof_team
rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_team -o t -- python3 tools/time_team.py 4096 16384 32768 > gpurun_out/r03/prof_team/run.log 2>&1
find gpurun_out/r03/prof_team -name "*kernel_stats*" | head
f=$(find gpurun_out/r03/prof_team -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: r["Name"]):
    if "pack" in r["Name"] or "elementwise" in r["Name"].lower(): continue
    print(f'{r["Name"][:100]:<100s} calls {r["Calls"]:>6s} avg_ns {float(r["AverageNs"]):>10.0f} min {r["MinNs"]:>8s} max {r["MaxNs"]:>8s}')
PY
