#!/bin/bash
# SQ counters of the compute-bound leg (correct_pixels, 16 marker slots, 65 536 filters): two rocprofv3 --pmc passes of
# `bench.py --only-pixels` (the program itself after `--`), digest -> gpurun_out/pmc_pixels/r03_pixels_sq.json (copy to profiles/).
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the GPU box exports GRAFT_REPO_ROOT; elsewhere: the script's repository
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
export OUT=gpurun_out/pmc_pixels
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --only-pixels > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 bench.py --only-pixels > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 bench.py --only-pixels > $OUT/p2.log 2>&1
timeout 120 python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, json, os
out = os.environ["OUT"]
acc = collections.defaultdict(lambda: [0.0, 0])
grid = 0
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(fs[-1])) if fs else []:
        if "correct_pixels_kernel" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        grid = int(r["Grid_Size"]) // 64
per = {k: v[0] / v[1] for k, v in acc.items()}
us = None
fs = sorted(glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])) if fs else []:
    if "correct_pixels_kernel" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
# GRBM_GUI_ACTIVE counts shader-clock cycles of the launch: clock = cycles / duration
mhz = (per.get("GRBM_GUI_ACTIVE", 0) / us) if us else None
d = {"kernel": "correct_pixels_kernel<float, 18> (left camera)", "batch": 65536, "marker_slots": 16, "waves": grid,
     "avg_launch_us_kernel_trace": us, "clock_MHz": mhz if mhz and 500 < mhz < 3500 else 2400.0,
     "SQ_INSTS_VALU_per_launch": per.get("SQ_INSTS_VALU"), "SQ_INSTS_VALU_per_wave": per.get("SQ_INSTS_VALU", 0) / max(grid, 1),
     "SQ_ACTIVE_INST_VALU_over_SQ_WAVE_CYCLES": (per.get("SQ_ACTIVE_INST_VALU", 0) / per["SQ_WAVE_CYCLES"]) if per.get("SQ_WAVE_CYCLES") else None,
     "counters_per_launch": per,
     "note": "VALU issue fraction of a launch = SQ_INSTS_VALU / (duration x 1024 SIMDs x clock / 4 cycles per wave64 VALU instruction); "
             "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (quad-cycle units both) is the same figure from the wave's own point of view (one wave per SIMD)"}
if us and d["SQ_INSTS_VALU_per_launch"]:
    d["valu_issue_frac_kernel_trace"] = d["SQ_INSTS_VALU_per_launch"] / (us * 1e-6 * 1024 * d["clock_MHz"] * 1e6 / 4)
json.dump(d, open(f"{out}/r03_pixels_sq.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
