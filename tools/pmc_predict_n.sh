#!/bin/bash
# SQ counters of the resident ImuUpdate loop (predict_n, tools/time_predict_n.py) for several library builds:
#   tools/pmc_predict_n.sh libA.so libB.so ...
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"      # the GPU box exports GRAFT_REPO_ROOT; elsewhere: the script's repository
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
mkdir -p gpurun_out/r05
for v in "$@"; do
  export FBUS_EKF_LIB=$PWD/$v
  export OUT=gpurun_out/r05/pmc_pn_$(basename $v .so)
  mkdir -p $OUT
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/time_predict_n.py 18 > $OUT/p1.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- python3 tools/time_predict_n.py 18 > $OUT/p2.log 2>&1
  python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
tot = collections.defaultdict(float); n = 0
for sub in ("p1", "p2"):
    fs = sorted(glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        if "predict_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
w = tot["SQ_WAVES"]
print(os.path.basename(out), " per wave (all launches K = 4..32 together): " + "  ".join(f"{k} {v / w:.0f}" for k, v in sorted(tot.items()) if k != "SQ_WAVES"))
if tot["SQ_INSTS_VALU"]:
    print("    ACTIVE_INST_VALU / INSTS_VALU = %.3f quad-cycles;  WAVE_CYCLES / INSTS_VALU = %.3f;  ACTIVE_INST_VALU / WAVE_CYCLES = %.3f;  WAIT_INST_ANY / WAVE_CYCLES = %.3f" % (
        tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_INSTS_VALU"], tot["SQ_WAVE_CYCLES"] / tot["SQ_INSTS_VALU"], tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"], tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"]))
PY
done 2>&1 | tee gpurun_out/r05/pmc_predict_n.txt
