#!/usr/bin/env python3
"""Static cost map of one kernel's ISA: instructions per source region, weighted by the issue costs measured with
scratch/ubench/ops.hip on gfx950 (fast VALU 1.0, slow VALU 1.7, transcendental 3.3, others counted separately).

    hipcc ... --cuda-device-only -S -g1 -o /tmp/rtx_g.s csrc/rtx.hip
    python scratch/isa_cost.py /tmp/rtx_g.s _Z13render_kernelILb0ELb1ELb0ELb0ELb1EEv12RenderParams
"""
import collections
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32",
        "v_lshlrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_mul_legacy_f32", "v_not_b32",
        "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}


def klass(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if not base.startswith("v_"):
        if base.startswith("s_"):
            return "salu"
        if base.startswith("ds_"):
            return "lds"
        if base.startswith(("global_", "flat_", "buffer_")):
            return "vmem"
        if base.startswith("scratch_"):
            return "spill"
        return "other"
    if base in ("v_readlane_b32", "v_writelane_b32"):
        return "lane"
    if base in QUARTER:
        return "quarter"
    if base in FAST:
        return "fast"
    return "slow"


def main():
    path, name = sys.argv[1], sys.argv[2]
    s = open(path).read()
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    files = {}
    for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    src = {f: open("/root/repo/ray-tracer_amd/csrc/" + f).read().split("\n") for f in ("rtx_device.hpp", "rtx.hip")}

    def region(f, l):
        if f == "rtx_device.hpp":
            for k in range(l - 1, 0, -1):
                m = re.match(r"^(?:template.*)?__device__.*?\b(\w+)\(", src[f][k])
                if m:
                    return "dev:" + m.group(1)
            return "dev:?"
        if f == "rtx.hip":
            text = src[f][l - 1].strip()
            return f"hip:{l:4d} {text[:70]}"
        return f

    cur = None
    table = collections.defaultdict(collections.Counter)
    order = []
    for line in s[i:j].split("\n"):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = line.strip()
        m = re.match(r"([a-z_0-9]+)\b", t)
        if not m or t.startswith((".", ";")) or t.endswith(":"):
            continue
        k = klass(m.group(1))
        if k == "other":
            continue
        r = region(*cur) if cur else "none"
        if r not in table:
            order.append(r)
        table[r][k] += 1
    W = {"fast": 1.0, "slow": 1.7, "quarter": 3.3, "lane": 1.7}
    print(f"{'valu cost':>9s} {'fast':>5s} {'slow':>5s} {'trans':>5s} {'lane':>5s} {'salu':>5s} {'lds':>4s} {'vmem':>4s} {'spill':>5s}  region")
    tot = collections.Counter()
    rows = []
    for r in order:
        c = table[r]
        cost = sum(W[k] * c[k] for k in W)
        rows.append((cost, r, c))
        tot.update(c)
    if "--sort" in sys.argv:
        rows.sort(key=lambda x: -x[0])
    for cost, r, c in rows:
        print(f"{cost:9.1f} {c['fast']:5d} {c['slow']:5d} {c['quarter']:5d} {c['lane']:5d} {c['salu']:5d} {c['lds']:4d} {c['vmem']:4d} {c['spill']:5d}  {r}")
    print(f"{sum(W[k] * tot[k] for k in W):9.1f} {tot['fast']:5d} {tot['slow']:5d} {tot['quarter']:5d} {tot['lane']:5d} {tot['salu']:5d} {tot['lds']:4d} {tot['vmem']:4d} {tot['spill']:5d}  TOTAL")


if __name__ == "__main__":
    main()
