#!/bin/bash
# C4 (998k triangles, 256 spp): leaf size / traversal cost of the SAH builder
for cfg in "RTX_SAH_MAXLEAF=4 RTX_SAH_CTRAV=1.0" "RTX_SAH_MAXLEAF=8 RTX_SAH_CTRAV=1.0" "RTX_SAH_MAXLEAF=8 RTX_SAH_CTRAV=2.0" "RTX_SAH_MAXLEAF=8 RTX_SAH_CTRAV=4.0" "RTX_SAH_MAXLEAF=16 RTX_SAH_CTRAV=4.0" "RTX_SAH_MAXLEAF=2 RTX_SAH_CTRAV=0.5" "RTX_SAH_MAXLEAF=4 RTX_SAH_CTRAV=2.0"; do
  r=$(env $cfg timeout -k 10 200 python -u bench.py --scene triangle-mesh --mesh-nlat 500 --spp 128 --steps 2 --warmup 1 --no-cpu 2>&1 | grep -o "\"value\": [0-9.]*\|box_tests_per_ray\": [0-9.]*\|prim_tests_per_ray\": [0-9.]*" | head -3 | tr '\n' ' ')
  echo "$cfg $r"
done
