#!/bin/bash
# usage: scratch/pmc.sh <tag> [bench args...]   -- separate rocprofv3 --pmc passes over one short bench run each
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
passes=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
 "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"
 "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_BRANCH"
 "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES"
)
i=0
for p in "${passes[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $p --output-format csv -d $out -o p$i -- python3 bench.py --no-cpu "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(float)
for f in sorted(glob.glob(out + "/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(f"{k:32s} {v:.6g}")
PY
