#!/bin/bash
# usage: scratch/ab_lib.sh <lib> <rounds> [bench.py arguments]  -- a workload with the in-tree library and with <lib>, alternating
# (default workload: the million-triangle configuration; prints Msamples/s, tests per ray, kernel time)
L=$1; N=${2:-3}; shift 2
ARGS=${*:---scene triangle-mesh --mesh-nlat 500 --spp 256}
for i in $(seq $N); do
  for lib in ray-tracer_amd/librtx.so $L; do
    v=$(RTX_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-secondary $ARGS 2>&1 | grep -o "\"value\": [0-9.]*\|box_tests_per_ray\": [0-9.]*\|prim_tests_per_ray\": [0-9.]*\|kernel_ms\": [0-9.]*" | head -4 | tr '\n' ' ') || exit 1
    echo "$lib $v"
  done
done
