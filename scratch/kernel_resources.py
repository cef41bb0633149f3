#!/usr/bin/env python3
"""Prints the register / scratch / LDS figures the compiler recorded for every kernel of librtx.so
(the code object's AMDGPU metadata note), so that what a profile reports can be checked against the build.

    python scratch/kernel_resources.py [path/to/librtx.so]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "librtx.so")
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        co = os.path.join(tmp, "gfx950.co")
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    rows = []
    for block in notes.split("- .agpr_count:")[1:]:
        get = lambda key: (re.search(rf"\.{key}:\s+(\S+)", block) or [None, "?"])[1]
        name = subprocess.check_output(["c++filt", get("name")], text=True).strip()
        rows.append((name.split("(")[0], get("vgpr_count"), get("sgpr_count"), get("private_segment_fixed_size"), get("group_segment_fixed_size"),
                     get("vgpr_spill_count"), get("sgpr_spill_count")))
    print(f"{'kernel':58s} {'vgpr':>5s} {'sgpr':>5s} {'scratch B':>9s} {'static LDS B':>12s} {'vgpr spills':>11s} {'sgpr spills':>11s}")
    for r in rows:
        print(f"{r[0]:58s} {r[1]:>5s} {r[2]:>5s} {r[3]:>9s} {r[4]:>12s} {r[5]:>11s} {r[6]:>11s}")


if __name__ == "__main__":
    main()
