#!/bin/bash
# C2 at 500 spp: scheduler constants (kernel_ms) after the scheduler pass became all scalar
for cfg in "6 8 3" "4 8 3" "8 8 3" "10 8 3" "6 6 3" "6 7 3" "6 10 3" "6 8 2" "6 8 4" "6 8 5" "8 7 3" "8 8 4" "5 8 3" "7 8 3"; do
  set -- $cfg
  v=$(RTX_SCHED_K=$1 RTX_W_PRIM=$2 RTX_W_SERV=$3 timeout -k 10 100 python -u bench.py --steps 3 --warmup 1 --no-cpu --spp 500 2>&1 | grep -o "\"kernel_ms\": [0-9.]*")
  echo "k=$1 w_prim=$2 w_serv=$3 $v"
done
