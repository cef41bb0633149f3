#!/bin/bash
# usage: scratch/regroup_bound.sh  -- how full would box steps be if finished lanes never waited?  The service weight is raised until a
# service pass starts as soon as any lane has finished (stage counters of the benchmark scene at 64 spp; -DRTX_STAGE_COUNTERS library)
for w in 3 8 24 64 512; do
  echo "# RTX_W_SERV=$w"
  RTX_W_SERV=$w RTX_LIB=$PWD/scratch/ab/librtx_sc.so timeout -k 10 200 python -u bench.py --steps 1 --warmup 0 --no-cpu --no-secondary --spp 64 2>&1 | grep -E "^\[stage|\"value\"" | head -3 | cut -c1-330
done
