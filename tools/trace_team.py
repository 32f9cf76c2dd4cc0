#!/usr/bin/env python3
"""per-kernel median durations of tools/time_team.py from a rocprofv3 kernel trace (csv):  python3 tools/trace_team.py DIR"""
import collections, csv, glob, os, statistics, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "_kernel<" not in n or any(k in n for k in ("pack_kernel", "reset_cov", "at::native")):
        continue
    name = n.replace("void (anonymous namespace)::", "").split("(")[0]
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 64)))
    grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
    agg[(grid // wg, name, wg // 64)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# frames_team_kernel serves the fused frame (F = 1) and the frame window (F = 8 here) alike: told apart by their duration
for key in [k for k in agg if k[1].startswith("frames_team_kernel")]:
    ds = agg.pop(key)
    cut = 3 * min(ds)
    for tag, part in (("  [1 frame]", [d for d in ds if d <= cut]), ("  [window of 8]", [d for d in ds if d > cut])):
        if part:
            agg[(key[0], key[1] + tag, key[2])] = part
print(f"{'tiles':>6} {'filters':>8}  {'kernel':<58} {'waves/tile':>10} {'launches':>8} {'median us':>10} {'min us':>8}")
for (tiles, name, roles), ds in sorted(agg.items()):
    print(f"{tiles:>6} {tiles * 64:>8}  {name[:58]:<58} {roles:>10} {len(ds):>8} {statistics.median(ds):>10.2f} {min(ds):>8.2f}")
