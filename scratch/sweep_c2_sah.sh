#!/bin/bash
# usage: scratch/sweep_c2_sah.sh  -- the SAH build's constants on the benchmark scene (array in LDS): leaf size, cost of a box test
# against a primitive test, leaves of one sphere without a box of their own.  Prints Msamples/s, tests per ray, kernel resources.
run() {
  env "$@" RTX_DEBUG=1 timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | python3 -c '
import sys, json
last = ""
for l in sys.stdin:
    if l.startswith("[rtx]"): last = l.strip()
    if l.startswith("{"):
        j = json.loads(l)
        print("   ", j["value"], "Msamples/s | box", j["box_tests_per_ray"], "prim", j["prim_tests_per_ray"], "|", last[21:90])
'
}
for bare in 0 1; do for leaf in 1 2 4; do for ct in 0.5 1 2; do
  echo "bare=$bare maxleaf=$leaf ctrav=$ct"
  run RTX_SAH_BARE_LEAF=$bare RTX_SAH_MAXLEAF=$leaf RTX_SAH_CTRAV=$ct RTX_LDS_ENTS_MAX=120000
done; done; done
