#!/bin/bash
# usage: scratch/stage_counters.sh [lib]  -- stage clocks and lanes per step of four scenes from a library built with -DRTX_STAGE_COUNTERS
L=${1:-scratch/ab/librtx_sc.so}
echo "# render_kernel built with -DRTX_STAGE_COUNTERS ($L), 1920x1080, 64 spp, depth 50: python bench.py <scene> --spp 64 --steps 1 --warmup 0 --no-cpu"
for s in "--scene bouncing-spheres" "--scene triangle-mesh --mesh-nlat 500" "--scene cornell-box" "--scene cornell-smoke" "--scene final-scene"; do
  echo "# $s --spp 64"
  RTX_LIB=$PWD/$L timeout -k 10 200 python -u bench.py --steps 1 --warmup 0 --no-cpu --no-secondary $s --spp 64 2>&1 | grep -E "^\[stage" | cut -c1-500 || exit 1
done
