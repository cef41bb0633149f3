#!/bin/bash
# usage: scratch/ab_libs.sh rounds libA libB ... [-- bench args]  -- alternate builds of librtx in one session, print Msamples/s
N=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in $(seq $N); do
  for L in "${libs[@]}"; do
    v=$(RTX_LIB=$PWD/$L timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"value\": [0-9.]*" | head -1)
    echo "[$L $*] $v"
  done
done
