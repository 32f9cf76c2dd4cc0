#!/bin/bash
mkdir -p gpurun_out/r04
python -m pytest tests/ -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04/gputests_mid.txt
./tools/_build/node_rate 1 2 4 8 2>&1 | tee gpurun_out/r04/node_rate.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_mid.json 2> gpurun_out/r04/bench_mid.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/bench_mid.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["avg_launch_us"], d["correct_kernel"]["avg_launch_us"], d["fused_frame"]["value"], d["fused_window"]["value"])
print({k:(v["value"], v["update_avg_launch_us"], v.get("valu_issue_frac")) for k,v in d["north_star_rows"].items() if isinstance(v,dict)})
print(d["legs_skipped"], d["fp64"]["value"], d["roofline"]["launch_policy"])
PY
