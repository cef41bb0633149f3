#!/bin/bash
for k in 4 6 8; do for wp in 8 10 13 16 20; do
  v=$(RTX_SCHED_K=$k RTX_W_PRIM=$wp timeout -k 10 100 python -u bench.py --steps 3 --warmup 1 --no-cpu --spp 500 2>&1 | grep -o "\"kernel_ms\": [0-9.]*")
  echo "k=$k w_prim=$wp $v"
done; done
