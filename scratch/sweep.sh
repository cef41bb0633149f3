#!/bin/bash
# usage: scratch/sweep.sh "<bench args>" VAR v1 v2 ...   -- bench once per value of one environment knob
args=$1; var=$2; shift 2
for v in "$@"; do
  r=$(env $var=$v timeout -k 10 200 python -u bench.py $args --steps 2 --warmup 1 --no-cpu 2>&1 | grep -o "\"value\": [0-9.]*")
  echo "$var=$v $r"
done
