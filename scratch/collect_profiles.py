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

# stage counters (scratch/stage_counters.sh > gpurun_out/stage_<tag>.txt): one pair of lines per scene (the bench renders twice)
stage = os.path.join(ROOT, "gpurun_out", f"stage_{tag}.txt")
if os.path.exists(stage):
    out, seen = [], 0
    for l in open(stage).read().split("\n"):
        if l.startswith("# --scene"):
            out.append(l)
            seen = 0
        elif l.startswith("[stage"):
            if seen < 2:
                out.append(l)
            seen += 1
        elif l.startswith("#"):
            out.append(l)
    note = """# the stage clocks include the clock reads themselves (each waits for outstanding memory reads), so they overstate the short stages;
# 'scheduling and item turnover' also holds the wait for the state word read that ends a primitive step
# (a library built with -DRTX_STAGE_COUNTERS keeps the compiled box loop of the packed LDS walk, which counts its steps; the product kernel runs
#  the hand-written pk_box_run_lds there, about a quarter cheaper per step: the box share of bouncing-spheres below is the compiled loop's)
# triangle-mesh runs the wide walk: its "box steps" are walk steps that serve lanes at a node (four child boxes each) and lanes at a
# primitive record alike, "tests/ray" there is lane-steps per ray (child-box tests per ray: 17.0, primitive tests per ray: 2.27 -- bench line)
# the packed binary walk, same command (profiles/r2_stage_counters.txt): triangle-mesh box 35.0% primitive 9.7% service 6.0% scheduling 49.4%,
#   box steps 2.04e8 at 23.2 lanes/step + prim steps 2.2e7 at 31.6 lanes/step, 5.0e7 scheduler passes
# cornell-box / cornell-smoke / final-scene: instances' primitives placed in the enclosing tree, frame-less variants at 6 waves per SIMD
#   (with transform brackets, RTX_PLACE_INSTANCES=0: cornell-box box 11.8% primitive 52.0%, 1.06e8 primitive steps at 33.3 lanes, 1.08e8 scheduler passes;
#    cornell-smoke 10.2% / 65.2%, 1.82e8 primitive steps at 28.5 lanes; final-scene 34.9% / 39.1%)
"""
    with open(os.path.join(dst, f"{tag}_stage_counters.txt"), "w") as g:
        g.write(out[0] + "\n" + note + "\n".join(out[1:]) + "\n")
# the render-walk gate's own log (python -m pytest tests -m gpu -s > gpurun_out/full_<tag>.log)
full = os.path.join(ROOT, "gpurun_out", f"full_{tag}.log")
if os.path.exists(full):
    import re
    lines = [re.sub(r"^tests\S* ", "", l) for l in open(full) if "[render walk]" in l]
    rays = sum(int(re.search(r"'rays': (\d+)", l).group(1)) for l in lines)
    clean = sum("'ids_differ': 0, 'near_ties': 0, 'edge_hits': 0, 'grazing_flips': 0, 't_bits_equal': 1.0, 'n_bits_equal': 1.0" in l for l in lines)
    with open(os.path.join(dst, f"{tag}_render_walk_gate.txt"), "w") as g:
        g.write("# tests/test_gpu_render_walk.py on MI355X (python -m pytest tests -m gpu -s): one line per comparison of rtx_trace(RTX_TRAVERSAL_RENDER)\n"
                f"# with the oracle / the exact walks; {len(lines)} comparisons, {rays} rays; {clean} of them with ids_differ 0, near_ties 0, edge_hits 0,\n"
                "# grazing_flips 0, t_bits_equal 1.0, n_bits_equal 1.0\n")
        g.writelines(lines)
