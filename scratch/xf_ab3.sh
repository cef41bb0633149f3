#!/bin/bash
# usage: scratch/xf_ab3.sh libA libB ...  -- instanced scenes, Msamples/s at 100 spp per library (scratch/ab/librtx_<name>.so)
for s in cornell-box cornell-smoke final-scene; do
  for rep in 1 2; do
    for L in "$@"; do
      v=$(RTX_LIB=$PWD/scratch/ab/librtx_$L.so timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --scene $s --spp 100 2>&1 | grep -o "\"value\": [0-9.]*" | head -1)
      echo "$s lib=$L $v"
    done
  done
done
