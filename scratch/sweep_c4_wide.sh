#!/bin/bash
# usage: scratch/sweep_c4_wide.sh  -- builder constants for the wide form on the million-triangle configuration (Msamples/s, tests per ray)
for e in "RTX_SAH_MAXLEAF=4 RTX_SAH_CTRAV=1" "RTX_SAH_MAXLEAF=2 RTX_SAH_CTRAV=1" "RTX_SAH_MAXLEAF=6 RTX_SAH_CTRAV=1" "RTX_SAH_MAXLEAF=4 RTX_SAH_CTRAV=0.5" "RTX_SAH_MAXLEAF=4 RTX_SAH_CTRAV=2" "RTX_SAH_MAXLEAF=6 RTX_SAH_CTRAV=2" "RTX_SAH_MAXLEAF=7 RTX_SAH_CTRAV=3" "RTX_SAH_MAXLEAF=2 RTX_SAH_CTRAV=0.5"; do
  v=$(env $e timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --scene triangle-mesh --mesh-nlat 500 --spp 256 2>&1 | grep -o "\"value\": [0-9.]*\|box_tests_per_ray\": [0-9.]*\|prim_tests_per_ray\": [0-9.]*" | head -3 | tr '\n' ' ')
  echo "$e $v"
done
