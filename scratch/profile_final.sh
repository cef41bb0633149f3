#!/bin/bash
# usage: scratch/profile_final.sh <tag>  -- kernel-trace stats of the default bench + HBM traffic passes
export TMPDIR=/tmp
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 bench.py --no-cpu --steps 5 --warmup 2 > $out/kt.log 2>&1 || { echo kt failed; tail -5 $out/kt.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o fetch -- python3 bench.py --no-cpu --steps 1 --warmup 0 > $out/fetch.log 2>&1 || { echo fetch failed; tail -5 $out/fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out -o write -- python3 bench.py --no-cpu --steps 1 --warmup 0 > $out/write.log 2>&1 || { echo write failed; tail -5 $out/write.log; exit 1; }
tail -1 $out/kt.log
cat $out/kt_kernel_stats.csv
grep -h render_kernel $out/fetch_counter_collection.csv $out/write_counter_collection.csv | awk -F'"' '{print $(NF-3), $(NF-2)}' 
