#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's device assembly (tooling, not product).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o dev.s plonky2_amd/csrc/p2hot.hip
       tools/isa_stats.py dev.s <substring of the mangled kernel name> [...]"""
import collections
import re
import sys

CHEAP = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
         "v_add_f32", "v_ashrrev_i32"}


def kernel_body(lines, sub):
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l and sub in l.split(":")[0]:
            start = i
            break
    if start is None:
        return None, None
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return lines[start].split(":")[0], body


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for sub in sys.argv[2:]:
        name, body = kernel_body(lines, sub)
        if body is None:
            print("not found:", sub)
            continue
        cnt = collections.Counter()
        for l in body:
            l = l.strip()
            if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
                continue
            m = re.match(r"([a-z_0-9]+)", l)
            if m:
                op = m.group(1)
                op = re.sub(r"_e(32|64)$", "", op)
                cnt[op] += 1
        valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
        cheap = sum(v for k, v in cnt.items() if k in CHEAP)
        mad = cnt["v_mad_u64_u32"] + cnt["v_mad_i64_i32"]
        ds = sum(v for k, v in cnt.items() if k.startswith("ds_"))
        gl = sum(v for k, v in cnt.items() if k.startswith("global_") or k.startswith("buffer_") or k.startswith("flat_"))
        sal = sum(v for k, v in cnt.items() if k.startswith("s_") and k not in ("s_nop", "s_waitcnt", "s_barrier"))
        print(f"{name}\n  VALU {valu} (cheap VOP2 {cheap}, mads {mad}, other {valu - cheap - mad})  DS {ds}  global {gl}  SALU {sal}  "
              f"s_nop {cnt['s_nop']}  s_waitcnt {cnt['s_waitcnt']}  s_barrier {cnt['s_barrier']}  scratch {sum(v for k, v in cnt.items() if k.startswith('scratch_'))}")
        top = ", ".join(f"{k} {v}" for k, v in cnt.most_common(24))
        print("  " + top)


main()
