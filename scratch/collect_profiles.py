#!/usr/bin/env python3
"""Copies the summaries of one scratch/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/<tag>_*:
kernel stats of the default bench command and of the million-triangle configuration, every dispatch of render_kernel with
its duration, the counter sums, the HBM counter rows bench.py reads, and the per-ray memory summary.

    python scratch/collect_profiles.py r3
"""
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
for cfg in ("c2", "c4"):
    shutil.copy(os.path.join(src, f"{cfg}_kt_kernel_stats.csv"), os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"))
    shutil.copy(os.path.join(src, f"{cfg}_pmc.txt"), os.path.join(dst, f"{tag}_{cfg}_pmc.txt"))
    if os.path.exists(os.path.join(src, f"{cfg}_memory_per_ray.json")):
        shutil.copy(os.path.join(src, f"{cfg}_memory_per_ray.json"), os.path.join(dst, f"{tag}_{cfg}_memory_per_ray.json"))
    # FETCH_SIZE and WRITE_SIZE rows of render_kernel (separate passes), as rocprofv3 wrote them
    rows, header = [], None
    for p in sorted(os.listdir(src)):
        if p.startswith(f"{cfg}_p") and p.endswith("_counter_collection.csv"):
            with open(os.path.join(src, p)) as f:
                r = csv.reader(f)
                h = next(r)
                header = header or h
                ci = h.index("Counter_Name")
                rows += [row for row in r if row[ci] in ("FETCH_SIZE", "WRITE_SIZE") and "render_kernel" in row[h.index("Kernel_Name")]]
    with open(os.path.join(dst, f"{tag}_{cfg}_hbm_counters.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(header)
        w.writerows(rows)
    # every dispatch of render_kernel in the kernel-trace pass
    with open(os.path.join(src, f"{cfg}_kt_kernel_trace.csv")) as f, open(os.path.join(dst, f"{tag}_{cfg}_kernel_launches.txt"), "w") as g:
        cmd = "python3 bench.py" if cfg == "c2" else "python3 bench.py --no-cpu --scene triangle-mesh --mesh-nlat 500 --spp 256"
        g.write(f"# every dispatch of render_kernel under `rocprofv3 --kernel-trace --stats -- {cmd}`, in launch order: variant, workgroups x threads, ms\n")
        if cfg == "c2":
            g.write("# (default command: warm-up + 3 timed frames + one rtx_render call of the headline workload, the 480x270 frame of the image-RMSE leg,\n"
                    "#  then the million-triangle block: warm-up + 3 timed frames and its 96x54 RMSE frame; the stats file averages over all launches of a variant,\n"
                    "#  small frames included -- the timed frames are the full-size ones listed here)\n")
        for r in csv.DictReader(f):
            if "render_kernel" in r["Kernel_Name"]:
                ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                g.write(f'{r["Kernel_Name"].replace("(RenderParams)", "")}  {int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])} x {r["Workgroup_Size_X"]}  {ms:.3f}\n')
print("\n".join(sorted(p for p in os.listdir(dst) if p.startswith(tag))))
