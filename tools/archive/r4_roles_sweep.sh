#!/bin/bash
# waves per tile of the measurement updates by batch size (the automatic choice: four up to a quarter of the chip's SIMDs in tiles, two up to half)
for b in 8192 16384 24576 32768 40960 49152; do
  for s in 4 16; do
    for r in 1 2 4; do
      echo -n "B $b slots $s roles $r: "
      python3 tools/run_pixels.py --batch $b --slots $s --roles $r --reps 10 2>&1 | grep correct_ | sed 's/.*left: \([0-9.]*\) us.*/pixels \1 us/' | tr '\n' ' '
      python3 tools/run_pixels.py --batch $b --slots $s --roles $r --reps 10 --corners 2>&1 | grep correct_ | sed 's/.*stereo: \([0-9.]*\) us.*/corners \1 us/'
    done
  done
done
