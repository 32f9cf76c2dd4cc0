mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q -s > gpurun_out/r03/pytest_gpu.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r03/pytest_gpu.log | tail -12
grep -E "config 5|correct_pixels" gpurun_out/r03/pytest_gpu.log | head -30
