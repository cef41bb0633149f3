#!/bin/bash
# usage: scratch/ab_env.sh "ENV_A" "ENV_B" [rounds] [bench args...]  -- alternate two environments in one session, print Msamples/s
A=$1; B=$2; N=${3:-2}; shift 3
for i in $(seq $N); do
  for E in "$A" "$B"; do
    v=$(env $E timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"value\": [0-9.]*\|box_tests_per_ray\": [0-9.]*")
    echo "[$E] $v"
  done
done
