#!/bin/bash
# C4 (998k triangles, 128 spp): scheduler constants for the HBM-resident walk (kernel_ms)
A="--scene triangle-mesh --mesh-nlat 500 --spp 128"
for cfg in "RTX_SCHED_K=6 RTX_W_PRIM=8 RTX_W_SERV=3" "RTX_SCHED_K=2 RTX_W_PRIM=8 RTX_W_SERV=3" "RTX_SCHED_K=12 RTX_W_PRIM=8 RTX_W_SERV=3" "RTX_SCHED_K=30 RTX_W_PRIM=8 RTX_W_SERV=3" \
           "RTX_SCHED_K=6 RTX_W_PRIM=4 RTX_W_SERV=3" "RTX_SCHED_K=6 RTX_W_PRIM=16 RTX_W_SERV=3" "RTX_SCHED_K=6 RTX_W_PRIM=8 RTX_W_SERV=1" "RTX_SCHED_K=6 RTX_W_PRIM=8 RTX_W_SERV=6" \
           "RTX_SCHED_K=6 RTX_W_PRIM=8 RTX_W_SERV=12" "RTX_SCHED_K=12 RTX_W_PRIM=16 RTX_W_SERV=6" "RTX_BLOCKS_PER_CU=1" "RTX_WAVES_PER_BLOCK=8"; do
  k=$(env $cfg timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu $A 2>&1 | grep -o "\"kernel_ms\": [0-9.]*")
  echo "$cfg $k"
done
