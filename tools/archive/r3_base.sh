mkdir -p gpurun_out/r03
python tools/run_configs.py > gpurun_out/r03/run_configs_base.txt 2>&1
for b in 4096 16384 32768 65536 73728; do
  python bench.py --batch $b --steps 10 --warmup 3 > gpurun_out/r03/bench_base_$b.json 2> gpurun_out/r03/bench_base_$b.err
done
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_base.log 2>&1
tail -3 gpurun_out/r03/pytest_base.log
cat gpurun_out/r03/run_configs_base.txt
