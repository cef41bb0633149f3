#!/bin/bash
# usage: scratch/sweep_sched.sh [bench args...]  -- scheduler constants of render_kernel, kernel_ms per setting
for k in 3 4 6; do for wp in 8 10 13; do for ws in 2 3 4; do
  v=$(RTX_SCHED_K=$k RTX_W_PRIM=$wp RTX_W_SERV=$ws timeout -k 10 100 python -u bench.py --steps 3 --warmup 1 --no-cpu "$@" 2>&1 | grep -o "\"kernel_ms\": [0-9.]*")
  echo "k=$k w_prim=$wp w_serv=$ws $v"
done; done; done
