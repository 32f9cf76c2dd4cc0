bash tools/profile_gpu.sh r03_b65536 "--batch 65536 --steps 20 --warmup 5" > /dev/null 2>&1
tail -22 gpurun_out/prof_r03_b65536/summary.txt
bash tools/profile_gpu.sh r03_b1048576 "--batch 1048576 --tile 16 --steps 3 --warmup 1" > /dev/null 2>&1
tail -22 gpurun_out/prof_r03_b1048576/summary.txt
bash tools/pmc_pixels.sh > /dev/null 2>&1
cat gpurun_out/pmc_pixels/summary.txt | head -40
