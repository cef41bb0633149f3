#!/bin/bash
# usage: scratch/all_scenes.sh [spp]  -- Msamples/s of every scene of the reference's zoo at 1920x1080 (bench.py --no-cpu, 2 timed steps)
spp=${1:-100}
for s in default bouncing-spheres checkered-spheres earth perlin-spheres quads simple-light cornell-box cornell-smoke final-scene; do
  r=$(timeout -k 10 300 python -u bench.py --scene $s --spp $spp --steps 2 --warmup 1 --no-cpu 2>&1 | grep -o "\"value\": [0-9.]*\|\"mrays_per_s\": [0-9.]*\|box_tests_per_ray\": [0-9.]*" | tr '\n' ' ')
  echo "$s spp=$spp $r"
done
