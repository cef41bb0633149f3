#!/bin/bash
# usage: scratch/sweep_c2_sched.sh  -- scheduler constants around the defaults on the default workload (Msamples/s, 2 timed frames each)
for e in "RTX_SCHED_K=6" "RTX_SCHED_K=4" "RTX_SCHED_K=5" "RTX_SCHED_K=8" "RTX_SCHED_K=10" "RTX_W_SERV=2" "RTX_W_SERV=4" "RTX_W_PRIM=6" "RTX_W_PRIM=10" "RTX_SCHED_K=6"; do
  v=$(env $e timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --no-secondary 2>&1 | grep -o "\"value\": [0-9.]*" | head -1)
  echo "$e $v"
done
