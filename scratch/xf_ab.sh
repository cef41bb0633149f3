#!/bin/bash
# usage: scratch/xf_ab.sh  -- instanced scenes: Msamples/s at 100 spp with instances' primitives placed in the enclosing tree (1) or behind brackets (0)
for s in cornell-box cornell-smoke final-scene; do
  for rep in 1 2; do
    for pl in 0 1; do
      v=$(RTX_PLACE_INSTANCES=$pl timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --scene $s --spp 100 2>&1 | grep -o "\"value\": [0-9.]*\|box_tests_per_ray\": [0-9.]*\|prim_tests_per_ray\": [0-9.]*" | head -3 | tr '\n' ' ')
      echo "$s place=$pl $v"
    done
  done
done
