#!/bin/bash
# usage: scratch/xf_ab.sh  -- instanced scenes: previous library against the new one at several RTX_XF_MIX, Msamples/s at 100 spp
for s in cornell-box cornell-smoke final-scene; do
  for rep in 1 2; do
    for cfg in "prev 0" "new 0" "new 2" "new 3" "new 5"; do
      set -- $cfg
      v=$(RTX_XF_MIX=$2 RTX_LIB=$PWD/scratch/ab/librtx_$1.so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --scene $s --spp 100 2>&1 | grep -o "\"value\": [0-9.]*" | head -1)
      echo "$s lib=$1 mix=$2 $v"
    done
  done
done
