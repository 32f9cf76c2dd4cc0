#!/bin/bash
# SQ counter pass for the hot kernels (run on the GPU box via gpurun).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sq_${1:-a}
mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-leg --kernel-timing off > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-leg --kernel-timing off > $OUT/p2.log 2>&1
timeout 120 python3 - <<'PY'
import csv, glob, collections, sys, os
out = os.environ.get("OUTDIR", "")
for sub in ("p1", "p2"):
    for f in glob.glob(f"gpurun_out/pmc_sq_*/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
            if "kernel" not in k: continue
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, d in acc.items():
            print(f, k)
            for c, (v, n) in d.items():
                print(f"   {c:<24} {v/n:16.1f} per launch")
PY
