#!/usr/bin/env python3
"""Copies what tools/profile_gpu.sh left under gpurun_out/prof_<TAG>/ into profiles/ under the names profiles/README.md
lists: <TAG>_kernel_stats.csv (rocprofv3 --stats, as written), <TAG>_pmc_hbm.csv (FETCH_SIZE / WRITE_SIZE: mean of the raw
counter per dispatch and kernel, KiB), <TAG>_summary.txt and rNN_digest_b<B>.json (tools/profile_digest.py).
Runs here, after the gpurun call:   python tools/profile_collect.py r02_b262144
gpurun_out/ keeps the run directories of earlier calls too; the newest file of each kind is the one taken."""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    if not f:
        raise SystemExit(f"nothing matches {pattern}")
    return f[-1]


def short(name):
    """'void (anonymous namespace)::correct_kernel<float, 18, 0, 0, true>(float*, ...)' -> 'correct_kernel<float, 18, 0, 0, true>'"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        depth += (ch == "<") - (ch == ">")
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(newest(src + "/trace/**/*_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, tag + "_summary.txt"))
    rnd, b = tag.split("_", 1)
    shutil.copy(os.path.join(src, "digest.json"), os.path.join(dst, f"{rnd}_digest_{b}.json"))
    rows = []
    for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(list)
        with open(newest(f"{src}/{sub}/**/*_counter_collection.csv")) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and not r["Kernel_Name"].startswith("__amd"):
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        rows += [(k, counter, len(v), sum(v) / len(v)) for k, v in sorted(acc.items())]
    with open(os.path.join(dst, tag + "_pmc_hbm.csv"), "w") as fh:
        fh.write("kernel,counter,dispatches,mean_value_KiB_per_dispatch\n")
        for k, c, n, m in rows:
            fh.write(f"\"{k}\",{c},{n},{m:.3f}\n")
    print("profiles/" + tag + "_{kernel_stats.csv,pmc_hbm.csv,summary.txt}", f"profiles/{rnd}_digest_{b}.json")


if __name__ == "__main__":
    main(sys.argv[1])
