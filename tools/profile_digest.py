#!/usr/bin/env python3
"""Digest of a tools/profile_gpu.sh run: per-kernel duration stats and per-launch HBM-side traffic.

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are collected in separate passes, are in KiB, and on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced streaming read (16 B/lane) -> doubled here; WRITE_SIZE is exact for
16 B/lane streaming stores."""
import csv, glob, os, sys, json
root = sys.argv[1]

def find(sub, pat):
    f = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return f[0] if f else None

out = {}
st = find("trace", "*kernel_stats.csv")
print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
if st:
    for r in csv.DictReader(open(st)):
        name = r["Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        print(f"{name:<46} calls {r['Calls']:>6}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}  {r['Percentage']:>6}%")
        out[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}

def pmc(sub, counter):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return {}
    acc = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        a = acc.setdefault(name, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
print("\n== HBM-side traffic per launch (separate --pmc passes; KiB counters; FETCH_SIZE x2 on gfx950) ==")
for name in sorted(set(fetch) | set(write)):
    if "kernel" not in name:
        continue
    fb = 2.0 * 1024.0 * fetch[name][0] / fetch[name][1] if name in fetch else float("nan")
    wb = 1024.0 * write[name][0] / write[name][1] if name in write else float("nan")
    us = out.get(name, {}).get("avg_us")
    rate = f"  -> {(fb + wb) / us / 1e6:6.2f} TB/s = {(fb + wb) / us / 1e6 / 8.0:5.3f} of 8 TB/s over the kernel-trace average" if us else ""
    print(f"{name:<46} fetch {fb/1e6:9.2f} MB  write {wb/1e6:9.2f} MB  total {(fb+wb)/1e6:9.2f} MB per launch{rate}")
    out.setdefault(name, {}).update({"fetch_bytes": fb, "write_bytes": wb})
for log in ("trace.log", "pmc_fetch.log", "pmc_write.log"):
    p = os.path.join(root, log)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            print(f"\n{log}: value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.4f}  predict avg (HIP events) "
                  f"{d['roofline']['avg_launch_us']:.2f} us  frac (bytes moved) {d['roofline']['frac']:.3f}  correct {d['correct_kernel']['avg_launch_us']:.2f} us "
                  f"frac {d['correct_kernel']['frac']:.3f}")
# the configuration the profile was taken on (bench.py quotes `traffic` from a digest only when it matches its own run)
cfg = {"batch": 65536, "dialect": "matlab", "markers": 4, "mode": "stacked", "args": sys.argv[2] if len(sys.argv) > 2 else ""}
a = cfg["args"].split()
for i, t in enumerate(a):
    if t == "--batch": cfg["batch"] = int(a[i + 1])
    if t == "--dialect": cfg["dialect"] = a[i + 1]
    if t == "--markers": cfg["markers"] = int(a[i + 1])
    if t == "--mode": cfg["mode"] = a[i + 1]
out["_config"] = cfg
json.dump(out, open(os.path.join(root, "digest.json"), "w"), indent=1)
