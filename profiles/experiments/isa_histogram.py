#!/usr/bin/env python3
"""Static instruction mix of one kernel in a gfx950 assembly listing (no GPU needed):

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S \\
          -Iinclude -Irpg_monocular_pose_estimator_amd/csrc rpg_monocular_pose_estimator_amd/csrc/mpe_k2.hip -o /tmp/mpe_k2.s
    python profiles/experiments/isa_histogram.py /tmp/mpe_k2.s 'k2_voteILb0ELb0ELi3' [more symbol substrings ...]

Counts are per instruction in the listing, NOT weighted by how often a block executes; they say what the code is made
of (how much of it is FP64 arithmetic, how much selects / moves / conversions around it), and the resource block says
what limits occupancy.  -> JSON on stdout."""
import json
import re
import sys

CLASSES = [
    ("fp64_fma", r"^v_fma_f64|^v_fmac_f64"),
    ("fp64_mul", r"^v_mul_f64"),
    ("fp64_add", r"^v_add_f64"),
    ("fp64_other", r"^v_\w+_f64|^v_cvt_f64|^v_cvt_\w+_f64"),   # rcp / rsq / sqrt / div_fixup / fmas / ldexp / cmp / cvt ...
    ("fp32", r"^v_\w+_f32|^v_pk_\w+_f32"),
    ("select_move", r"^v_cndmask|^v_mov_b|^v_accvgpr|^v_readlane|^v_readfirstlane|^v_writelane|^v_swap|^v_permlane|^v_bfrev"),
    ("int_valu", r"^v_"),                                          # whatever VALU is left: integer / bit / compare
    ("lds", r"^ds_"),
    ("vmem", r"^global_|^buffer_|^flat_|^scratch_"),
    ("smem", r"^s_load|^s_buffer_load|^s_store"),
    ("wait_barrier", r"^s_waitcnt|^s_barrier|^s_nop|^s_sleep"),
    ("branch", r"^s_cbranch|^s_branch|^s_setpc|^s_swappc|^s_endpgm"),
    ("salu", r"^s_"),
]
RES = ("next_free_vgpr", "next_free_sgpr", "accum_offset", "group_segment_fixed_size", "private_segment_fixed_size")


def kernel_body(lines, sub):
    start = None
    for i, ln in enumerate(lines):
        if start is None and ln.startswith("_Z") and sub in ln and ln.rstrip().split(":")[0].startswith("_Z"):
            start = i
            name = ln.split(":")[0]
        elif start is not None and ln.startswith(".Lfunc_end"):
            return name, lines[start:i]
    raise KeyError(sub)


def resources(lines, name):
    out, on = {}, False
    for ln in lines:
        if ln.strip().startswith(".amdhsa_kernel") and name in ln:
            on = True
        elif on and ln.strip().startswith(".end_amdhsa_kernel"):
            break
        elif on:
            m = re.match(r"\s*\.amdhsa_(\w+)\s+(\S+)", ln)
            if m and m.group(1) in RES:
                out[m.group(1)] = int(m.group(2), 0)
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    report = {}
    for sub in sys.argv[2:]:
        name, body = kernel_body(lines, sub)
        hist = {c: 0 for c, _ in CLASSES}
        top = {}
        for ln in body:
            t = ln.strip()
            if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
                continue
            op = t.split()[0]
            for c, pat in CLASSES:
                if re.match(pat, op):
                    hist[c] += 1
                    break
            else:
                hist.setdefault("other", 0)
                hist["other"] += 1
            top[op] = top.get(op, 0) + 1
        valu = sum(hist[c] for c in ("fp64_fma", "fp64_mul", "fp64_add", "fp64_other", "fp32", "select_move", "int_valu"))
        report[name] = {"instructions": sum(hist.values()), "valu": valu, "classes": hist,
                        "valu_share": {c: round(hist[c] / valu, 3) for c in ("fp64_fma", "fp64_mul", "fp64_add", "fp64_other",
                                                                              "fp32", "select_move", "int_valu")},
                        "top_opcodes": sorted(top.items(), key=lambda kv: -kv[1])[:16], "resources": resources(lines, name)}
    json.dump(report, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
