#!/bin/bash
# usage: scratch/profile_round.sh <tag>  -- the profiles kept under profiles/: for the default bench (C2) and for the
# million-triangle configuration (C4): rocprofv3 kernel-trace stats, instruction / LDS / wait counters, HBM traffic
# (FETCH_SIZE and WRITE_SIZE in passes of their own), L2 hit counters.  Every pass is its own short run.
export TMPDIR=/tmp
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
C4="--scene triangle-mesh --mesh-nlat 500 --spp 256"
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 "$@" > $out/$name.log 2>&1 || { echo "$name failed"; tail -5 $out/$name.log; exit 1; }
}
# kernel-trace stats of the default bench command itself, and of the C4 configuration
run c2_kt --kernel-trace --stats --output-format csv -d $out -o c2_kt -- python3 bench.py
run c4_kt --kernel-trace --stats --output-format csv -d $out -o c4_kt -- python3 bench.py --no-cpu $C4
i=0
for p in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
         "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  run c2_p$i --pmc $p --output-format csv -d $out -o c2_p$i -- python3 bench.py --no-cpu --steps 1 --warmup 0
  run c4_p$i --pmc $p --output-format csv -d $out -o c4_p$i -- python3 bench.py --no-cpu --steps 1 --warmup 0 $C4
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for cfg in ("c2", "c4"):
    agg = collections.defaultdict(float)
    calls = collections.Counter()
    for f in sorted(glob.glob(f"{out}/{cfg}_p*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                calls[r["Counter_Name"]] += 1
    with open(f"{out}/{cfg}_pmc.txt", "w") as g:
        g.write(f"# render_kernel, summed over the launches of one `bench.py --no-cpu --steps 1 --warmup 0` run per pass (launches per pass in the last column)\n")
        for k, v in sorted(agg.items()):
            g.write(f"{k:32s} {v:.6g} {calls[k]}\n")
    print(open(f"{out}/{cfg}_pmc.txt").read())
    # where the walk's bytes came from, per ray (what bench.py quotes under roofline.measured_per_ray)
    import json, re
    rays = None
    for line in open(f"{out}/{cfg}_p0.log"):
        if line.startswith("{"):
            j = json.loads(line)
            w, h, spp = re.search(r"(\d+)x(\d+), (\d+) spp", j["config"]["workload"]).groups()
            rays = j["rays_per_sample"] * int(w) * int(h) * int(spp)
    if rays and calls.get("TCC_REQ_sum"):
        per = lambda k: agg[k] / calls[k] / rays
        summary = {
            "note": "render_kernel, per launch and ray, from rocprofv3 PMC passes of `bench.py --no-cpu --steps 1 --warmup 0` on this workload "
                    "(scratch/profile_round.sh); FETCH_SIZE is doubled as the gfx950 guide prescribes (it tallies 128-byte requests at 64 bytes)",
            "rays_per_launch": rays,
            "vector_l1_accesses_per_ray": round(per("TCP_TOTAL_CACHE_ACCESSES_sum"), 2) if calls.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
            "l2_requests_per_ray": round(per("TCC_REQ_sum"), 3),
            "l2_request_bytes_per_ray_at_128": round(128 * per("TCC_REQ_sum"), 1),
            "l2_hit_rate": round(agg["TCC_HIT_sum"] / (agg["TCC_HIT_sum"] + agg["TCC_MISS_sum"]), 4),
            "bytes_past_l2_per_ray": round(2 * 1024 * per("FETCH_SIZE"), 1) if calls.get("FETCH_SIZE") else None,
            "waiting_share_of_wave_time": round(agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"], 4) if calls.get("SQ_WAIT_ANY") else None,
        }
        json.dump(summary, open(f"{out}/{cfg}_memory_per_ray.json", "w"), indent=1)
        print(json.dumps(summary, indent=1))
PY
grep -h render_kernel $out/c2_kt_kernel_stats.csv $out/c4_kt_kernel_stats.csv
