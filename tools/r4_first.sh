#!/bin/bash
# round 4, first call: instruction issue costs + this round's baseline numbers on this box
mkdir -p gpurun_out/r04
./tools/_build/exp_issue_rates > gpurun_out/r04/issue_rates.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err
python -m pytest tests/test_pixels_gpu.py tests/test_vision_gpu.py -x -q -m gpu -s 2>&1 | grep -E "parity|perf|passed|failed|Error" > gpurun_out/r04/pixels_base.txt
tail -3 gpurun_out/r04/pixels_base.txt
