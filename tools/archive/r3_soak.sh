# the team-kernel tests N times in a row (intermittent races would show as a failure in some round)
N=${1:-15}
fail=0
for i in $(seq 1 $N); do
  python -m pytest tests/test_team_gpu.py tests/test_vision_gpu.py tests/test_pixels_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -1 | grep -q "passed" || { fail=$((fail+1)); echo "round $i FAILED"; }
done
echo "soak: $N rounds, $fail failed"
