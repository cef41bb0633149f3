#!/bin/bash
# usage: scratch/sweep_env.sh VAR "v1 v2 ..." [bench args...]  -- kernel_ms of the bench per value of one environment knob
var=$1; vals=$2; shift 2
for r in 1 2; do for v in $vals; do
  k=$(env $var=$v timeout -k 10 100 python -u bench.py --steps 3 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"kernel_ms\": [0-9.]*")
  echo "$var=$v $k"
done; done
