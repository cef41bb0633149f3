#!/bin/bash
# A/B of pad slots in the LDS-resident packed array (RTX_PK_PAD = a pad behind every n-th sphere, 0 = none): frame rate of the
# default bench command, then LDS counters per setting.
export TMPDIR=/tmp
bash scratch/ab3.sh 2 RTX_PK_PAD=0 RTX_PK_PAD=2 RTX_PK_PAD=4 RTX_PK_PAD=8 -- --no-secondary | grep -v "^\[--"
for p in 0 4; do
  out=gpurun_out/pmc_pad$p
  RTX_PK_PAD=$p timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $out -o x -- python3 bench.py --no-cpu --no-secondary --steps 1 --warmup 0 > $out.log 2>&1 || { echo "pmc pass failed"; tail -3 $out.log; exit 1; }
  python3 - $out $p <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print("RTX_PK_PAD=" + sys.argv[2], {k: f"{v:.4g}" for k, v in sorted(agg.items())})
PY
done
