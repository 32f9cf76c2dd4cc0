#!/bin/bash
# round 4: corner-row update -- launch times and the tests that gate it (the round-3 kernels it replaced: profiles/r04_corners_times.txt)
mkdir -p gpurun_out/r04
{
for leg in ""; do
  echo "== ${leg:-round-4 kernels}"
  env $leg python3 tools/run_pixels.py --corners 2>&1 | grep correct_
  env $leg python3 tools/run_pixels.py --corners --batch 16384 --slots 4 2>&1 | grep correct_
done
} | tee gpurun_out/r04/corners_times.txt
python -m pytest tests/test_vision_gpu.py tests/test_configs_gpu.py -q -m gpu -s -k "corners or config3 or config5 or config_5 or config_3" 2>&1 | grep -E "parity|perf|passed|failed|Error|assert" | tee gpurun_out/r04/corners_tests.txt | tail -40
