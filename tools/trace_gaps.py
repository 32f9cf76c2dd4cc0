#!/usr/bin/env python3
"""Launch gaps along a rocprofv3 --kernel-trace run of bench.py: for consecutive kernels of the library (sorted by start),
the idle time between the end of one and the start of the next, and the kernel durations, averaged over windows of W
launches -- shows whether a long timed region slows down (clock / power management) or the host falls behind.
  python tools/trace_gaps.py <dir with *_kernel_trace.csv> [W]"""
import csv, glob, os, sys
d = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 460
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "predict_kernel" in n or "correct_kernel" in n or "frame_kernel" in n:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "p" if "predict" in n else ("c" if "correct" in n else "f")))
rows.sort()
print(f"{len(rows)} launches in {f}")
print(f"{'launch':>8} {'t ms':>8} {'predict us':>10} {'correct us':>10} {'frame us':>9} {'gap us':>7} {'gaps>5us':>8} {'busy':>6}")
for lo in range(0, len(rows) - 1, W):
    win = rows[lo:lo + W + 1]
    gaps = [max(0, win[i + 1][0] - win[i][1]) / 1e3 for i in range(len(win) - 1)]
    gaps_small = [g for g in gaps if g < 1000]
    dur = {k: [(e - s) / 1e3 for s, e, kk in win[:-1] if kk == k] for k in "pcf"}
    avg = lambda v: sum(v) / len(v) if v else float("nan")
    span = (win[-1][0] - win[0][0]) / 1e3
    busy = sum((e - s) / 1e3 for s, e, _ in win[:-1]) / span if span else 0
    print(f"{lo:>8} {(win[0][0] - rows[0][0]) / 1e6:>8.2f} {avg(dur['p']):>10.2f} {avg(dur['c']):>10.2f} {avg(dur['f']):>9.2f} "
          f"{avg(gaps_small):>7.2f} {sum(g > 5 for g in gaps_small):>8} {busy:>6.3f}")
big = [(i, max(0, rows[i + 1][0] - rows[i][1]) / 1e3) for i in range(len(rows) - 1)]
big = [(i, g) for i, g in big if 30 < g < 5000]
print("gaps of 30 us .. 5 ms (launch index: us):", " ".join(f"{i}:{g:.0f}" for i, g in big))
