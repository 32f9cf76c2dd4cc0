#!/bin/bash
# (any round; R=$2, default 06) the reprojection-row update at 65 536 filters x 16 marker slots -- launch times (the round-3 kernel it replaced: profiles/r04_pixels_times.txt),
# kernel trace and SQ counters (own passes: --pmc never together with other trace domains).  TAG=$1 names the output directory;
# $3 (optional): extra arguments of tools/run_pixels.py for the profiled passes, e.g. --stereo (the digest then describes that launch).
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
TAG=${1:-a}
R=${2:-06}
export XARGS="${3:-}"
export OUT=gpurun_out/r${R}/pix_$TAG
mkdir -p $OUT
python3 tools/run_pixels.py --both > $OUT/times.txt 2>&1

python3 tools/run_pixels.py --both --batch 16384 --slots 4 >> $OUT/times.txt 2>&1
python3 bench.py --only-pixels > $OUT/north_star_rows.json 2> $OUT/north_star_rows.err
cat $OUT/times.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/run_pixels.py $XARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/run_pixels.py $XARGS > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -- python3 tools/run_pixels.py $XARGS > $OUT/p2.log 2>&1
timeout 120 python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, json, os
out = os.environ["OUT"]
acc = collections.defaultdict(lambda: [0.0, 0])
grid = 0
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(fs[-1])) if fs else []:
        if "correct_pixels2_kernel" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        grid = int(r["Grid_Size"]) // 64
per = {k: v[0] / v[1] for k, v in acc.items()}
us = None
name = None
# the kernel's time: the MEDIAN of its dispatches in the kernel trace (the first dispatch of a process includes the code load: 18 ms
# in a run of 14 -- the --stats average is useless for so few calls)
ft = sorted(glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
dur = []
for r in csv.DictReader(open(ft[-1])) if ft else []:
    if "correct_pixels2_kernel" in r["Kernel_Name"]:
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3); name = r["Kernel_Name"]
if dur:
    dur.sort(); us = dur[len(dur) // 2]
fs = sorted(glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
import shutil
if fs: shutil.copy(fs[-1], f"{out}/kernel_stats.csv")
mhz = (per.get("GRBM_GUI_ACTIVE", 0) / 8 / us) if us else None      # GRBM_GUI_ACTIVE sums the 8 XCDs
if name is None:
    raise SystemExit("no correct_pixels row in the kernel trace: nothing to summarise")
d = {"kernel": name.replace("void (anonymous namespace)::", "").split("(")[0], "batch": 65536, "marker_slots": 16, "camera": "stereo" if "--stereo" in os.environ.get("XARGS", "") else "left", "waves": grid, "simds": 1024,
     "avg_launch_us_kernel_trace": us, "launch_us_is": "median of the kernel-trace dispatches", "clock_MHz": mhz,
     "SQ_INSTS_VALU_per_wave": per.get("SQ_INSTS_VALU", 0) / max(grid, 1), "counters_per_launch": per}
if us and per.get("SQ_INSTS_VALU") and mhz:
    d["valu_issue_frac_kernel_trace"] = per["SQ_INSTS_VALU"] / (us * 1e-6 * 1024 * mhz * 1e6 / 4)
    d["SQ_ACTIVE_INST_VALU_over_SQ_WAVE_CYCLES"] = per.get("SQ_ACTIVE_INST_VALU", 0) / per["SQ_WAVE_CYCLES"]
    d["SQ_WAIT_ANY_over_SQ_WAVE_CYCLES"] = per.get("SQ_WAIT_ANY", 0) / per["SQ_WAVE_CYCLES"]
    d["SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES"] = per.get("SQ_WAIT_INST_ANY", 0) / per["SQ_WAVE_CYCLES"]
json.dump(d, open(f"{out}/pixels_sq.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
