mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_team_gpu.py -x -q -s > gpurun_out/r03/team_tests.log 2>&1
tail -25 gpurun_out/r03/team_tests.log
