#!/usr/bin/env python3
"""tools/run_configs.py under `rocprofv3 --kernel-trace`: the same table with the KERNEL-TRACE duration of every case beside the
HIP-event figure (a HIP-event bracket around a single launch reads 2-3 us long, which is 30-50 % of a 5-10 us kernel).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/run_configs.py      (FBUS_RUN_CONFIGS_JSON=DIR/cases.json)
    python3 tools/run_configs_rocprof.py DIR
Each case of run_configs.py is one contiguous run of launches of ONE library kernel; the trace is cut into such runs in time order."""
import csv, glob, json, os, statistics, sys
root = sys.argv[1]
cases = json.load(open(os.path.join(root, "cases.json")))
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    n = r["Kernel_Name"]
    if "_kernel<" not in n or any(k in n for k in ("pack_kernel", "reset_cov", "at::native", "marker_pose")):
        continue
    key = (n.replace("void (anonymous namespace)::", "").split("(")[0], int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append([key, [d]])
runs = [r for r in runs if len(r[1]) >= 10]
print("config  case                                             kernel (grid)                                                  n    us median   us HIP-event   algorithmic GB/s   frac of 8 TB/s")
if len(runs) != len(cases):
    print(f"!! {len(runs)} kernel runs for {len(cases)} cases", file=sys.stderr)
for c, (key, ds) in zip(cases, runs):
    ds = ds[3:] if len(ds) > 6 else ds               # the warm-up launches of the case
    us = statistics.median(ds)
    gbs = c["bytes_per_launch"] / us / 1e3
    print(f"{c['config']:>4}    {c['desc']:<48} {key[0][:52]:<52} ({key[1]:>7d}) {len(ds):>5d} {us:10.2f} {c['us_hip_events']:12.2f} {gbs:16.0f} {gbs / 8000:14.2f}")
