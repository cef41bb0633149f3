#!/bin/bash
# C4 (998k-triangle mesh, entries in HBM/L2): kernel time + L2 hit rate + HBM traffic, separate passes
export TMPDIR=/tmp
out=gpurun_out/c4
mkdir -p $out
A="--scene triangle-mesh --mesh-nlat 500 --spp 64 --steps 1 --warmup 0 --no-cpu"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 bench.py $A > $out/kt.log 2>&1 || { echo kt failed; tail -5 $out/kt.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out -o tcc -- python3 bench.py $A > $out/tcc.log 2>&1 || { echo tcc failed; tail -5 $out/tcc.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python3 bench.py $A > $out/fetch.log 2>&1 || { echo fetch failed; tail -5 $out/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $out -o sq -- python3 bench.py $A > $out/sq.log 2>&1 || { echo sq failed; tail -5 $out/sq.log; exit 1; }
tail -1 $out/kt.log
head -3 $out/kt_kernel_stats.csv
grep -h render_kernel $out/tcc_counter_collection.csv $out/fetch_counter_collection.csv $out/sq_counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}'
