#!/bin/bash
# usage: scratch/c4_stages.sh [ENV=VALUE ...]  -- stage counters of the million-triangle configuration (build with -DRTX_STAGE_COUNTERS)
env "$@" RTX_LIB=$PWD/scratch/ab/librtx_sc.so timeout -k 10 200 python -u bench.py --steps 1 --warmup 0 --no-cpu --scene triangle-mesh --mesh-nlat 500 --spp 256 2>&1 | grep -E "^\[stage|\"value\"" | cut -c1-400
