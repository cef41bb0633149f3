#!/bin/bash
# usage: scratch/c4_quick.sh [ENV=VALUE ...]  -- one short run of the million-triangle configuration, prints the line's key figures
env "$@" RTX_DEBUG=1 timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --scene triangle-mesh --mesh-nlat 500 --spp 256 2>&1 | python3 -c '
import sys, json
for l in sys.stdin:
    if l.startswith("[rtx]"): last = l.strip()
    if l.startswith("{"):
        j = json.loads(l)
        print(last)
        print({k: j[k] for k in ("value", "mrays_per_s", "rays_per_sample", "box_tests_per_ray", "prim_tests_per_ray", "kernel_ms")})
'
