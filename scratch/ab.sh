#!/bin/bash
# usage: scratch/ab.sh libA libB [rounds]   -- alternate two builds of librtx in one session, print Msamples/s
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    v=$(RTX_LIB=$L timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | grep -o "\"value\": [0-9.]*")
    echo "$L $v"
  done
done
