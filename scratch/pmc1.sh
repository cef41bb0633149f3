#!/bin/bash
# usage: scratch/pmc1.sh <tag> <lib>  -- one rocprofv3 --pmc pass (instruction counts) over a short bench run with the given library
export TMPDIR=/tmp
tag=$1; lib=$2
out=gpurun_out/pmc1_$tag
mkdir -p $out
export RTX_LIB=$lib
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $out -o p -- python3 bench.py --no-cpu --spp 128 --steps 1 --warmup 0 > $out/p.log 2>&1 || { echo "pass failed"; tail -5 $out/p.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in sorted(glob.glob(sys.argv[1] + "/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(" ".join(f"{k}={v:.4g}" for k, v in sorted(agg.items())))
PY
