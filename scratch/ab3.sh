#!/bin/bash
# usage: scratch/ab3.sh rounds "ENV_1" "ENV_2" ... [-- bench args]  -- alternate environments in one session, print Msamples/s
N=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in $(seq $N); do
  for E in "${envs[@]}"; do
    v=$(env $E timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"value\": [0-9.]*" | head -1)
    echo "[$E] $v"
  done
done
