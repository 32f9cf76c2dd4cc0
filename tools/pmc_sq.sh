#!/bin/bash
# SQ counter passes for the hot kernels (run on the GPU box via gpurun; the program itself after `--`).
#   tools/pmc_sq.sh TAG ["extra bench.py arguments"]      e.g.  tools/pmc_sq.sh b262144 "--batch 262144"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the GPU box exports GRAFT_REPO_ROOT; elsewhere: the script's repository
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
TAG=${1:-a}
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-hbm-leg --no-extra-legs --kernel-timing off ${2:-}"
export OUT=gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
timeout 120 python3 - <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, os
out = os.environ["OUT"]
print(f"SQ counters ({out}; quad-cycle units for *_CYCLES / WAIT / ACTIVE); per launch and per wave")
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    grid = {}
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")
        depth = 0
        for i, ch in enumerate(k):
            depth += (ch == "<") - (ch == ">")
            if ch == "(" and depth == 0:
                k = k[:i]; break
        if "kernel" not in k or k.startswith("__amd"):
            continue
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        grid[k] = int(r["Grid_Size"]) // 64
    for k, d in sorted(acc.items()):
        print(f"{k}   [{sub}]  {grid[k]} waves")
        for c, (v, n) in sorted(d.items()):
            print(f"   {c:<24} {v / n:16.1f} per launch {v / n / grid[k]:12.1f} per wave")
PY
