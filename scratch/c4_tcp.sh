#!/bin/bash
# usage: scratch/c4_tcp.sh tag [ENV=VALUE ...]  -- texture-address / vector-L1 counters of the million-triangle configuration (64 spp)
export TMPDIR=/tmp
tag=$1; shift
for e in "$@"; do export "$e"; done
out=gpurun_out/c4tcp_$tag
mkdir -p $out
A="--scene triangle-mesh --mesh-nlat 500 --spp 64 --steps 1 --warmup 0 --no-cpu"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE --output-format csv -d $out -o p -- python3 bench.py $A > $out/p.log 2>&1 || { echo failed; tail -5 $out/p.log; exit 1; }
echo "== $tag $*"
grep -h render_kernel $out/p_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}'
