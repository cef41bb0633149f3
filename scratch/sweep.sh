#!/bin/bash
# usage: scratch/sweep.sh VAR v1 v2 ...   -- bench once per value of one environment knob
var=$1; shift
for v in "$@"; do
  r=$(env $var=$v timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep -o "\"value\": [0-9.]*")
  echo "$var=$v $r"
done
