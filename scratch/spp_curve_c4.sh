#!/bin/bash
# C4 (998k triangles): kernel time against samples per pixel.  usage: scratch/spp_curve_c4.sh [extra bench args]
for spp in 16 64 256; do
  r=$(RTX_DEBUG=1 timeout -k 10 300 python -u bench.py --scene triangle-mesh --mesh-nlat 500 --spp $spp --steps 2 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"value\": [0-9.]*\|\"kernel_ms\": [0-9.]*\|rays_per_sample\": [0-9.]*\|[0-9]* batches of [0-9]* spp, [0-9]* items" | sort -u | tr '\n' ' ')
  echo "spp=$spp $r"
done
