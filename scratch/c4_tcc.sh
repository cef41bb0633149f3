#!/bin/bash
# usage: scratch/c4_tcc.sh tag [ENV=VALUE ...]  -- L2 requests / hits / misses of the million-triangle configuration (64 spp)
export TMPDIR=/tmp
tag=$1; shift
for e in "$@"; do export "$e"; done
out=gpurun_out/c4tcc_$tag
mkdir -p $out
A="--scene triangle-mesh --mesh-nlat 500 --spp 64 --steps 1 --warmup 0 --no-cpu"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out -o p -- python3 bench.py $A > $out/p.log 2>&1 || { echo failed; tail -5 $out/p.log; exit 1; }
echo "== $tag $*"
grep -h render_kernel $out/p_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}'
