#!/bin/bash
# usage: scratch/depth_curve.sh [bench args]  -- kernel time and ray counts against the depth limit: what the deep bounces cost
for d in 1 2 3 5 10 20 50; do
  r=$(timeout -k 10 300 python -u bench.py --depth $d --steps 2 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"kernel_ms\": [0-9.]*\|rays_per_sample\": [0-9.]*\|box_tests_per_ray\": [0-9.]*\|prim_tests_per_ray\": [0-9.]*" | sort -u | tr '\n' ' ')
  echo "depth=$d $r"
done
