#!/bin/bash
# usage: scratch/spp_small.sh [bench args]  -- kernel time at 1..8 samples per pixel and two depth limits: the fixed part of a launch
for d in 50 5; do for spp in 1 4 8; do
  r=$(timeout -k 10 200 python -u bench.py --spp $spp --depth $d --steps 4 --warmup 2 --no-cpu "$@" 2>&1 | grep -o "\"kernel_ms\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | head -3 | tr '\n' ' ')
  echo "depth=$d spp=$spp $r"
done; done
