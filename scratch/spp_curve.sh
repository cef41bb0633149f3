#!/bin/bash
# usage: scratch/spp_curve.sh [bench args]  -- kernel time against samples per pixel: the fixed part of a launch
for spp in 8 16 32 64 125 250 500 1000; do
  r=$(timeout -k 10 200 python -u bench.py --spp $spp --steps 4 --warmup 2 --no-cpu "$@" 2>&1 | grep -o "\"value\": [0-9.]*\|\"kernel_ms\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | head -3 | tr '\n' ' ')
  echo "spp=$spp $r"
done
