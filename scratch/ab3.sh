#!/bin/bash
# usage: scratch/ab3.sh rounds "ENV LIB" "ENV LIB" ...  -- alternate (environment, library) pairs in one session; prints kernel_ms
N=$1; shift
for i in $(seq $N); do
  for spec in "$@"; do
    k=$(env $spec timeout -k 10 100 python -u bench.py --steps 3 --warmup 1 --no-cpu --spp 500 2>&1 | grep -o "\"kernel_ms\": [0-9.]*")
    echo "[$spec] $k"
  done
done
