#!/usr/bin/env python3
"""Per-position kernel durations from a rocprofv3 --kernel-trace CSV: predict launches by their position after the
last correct, correct launches, and the idle gap in front of each.  usage: trace_positions.py <dir or csv>"""
import collections, csv, glob, os, statistics as st, sys

def main(path):
    f = path if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    cnt, prev_end, prev_kind = 0, None, None
    for r in rows:
        n = r["Kernel_Name"]
        kind = "P" if "predict_kernel" in n else "C" if "correct_kernel" in n else None
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if kind is None:
            prev_end, prev_kind = None, None
            continue
        key = f"predict #{cnt}" if kind == "P" else "correct"
        dur[key].append((e - s) / 1e3)
        if prev_end is not None:
            gap[key].append((s - prev_end) / 1e3)
        cnt = cnt + 1 if kind == "P" else 0
        prev_end, prev_kind = e, kind
    tot = 0.0
    for k in sorted(dur):
        g = st.median(gap[k]) if gap[k] else float("nan")
        print(f"{k:12s} n {len(dur[k]):4d}  duration median {st.median(dur[k]):6.2f} mean {st.mean(dur[k]):6.2f} us   idle before it {g:5.2f} us")
    frame = [st.median(dur[k]) + (st.median(gap[k]) if gap[k] else 0) for k in dur]
    print(f"sum of medians (one frame of {len(dur) - 1} predicts + correct, gaps included): {sum(frame):.1f} us")

if __name__ == "__main__":
    main(sys.argv[1])
