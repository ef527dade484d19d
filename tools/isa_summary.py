#!/usr/bin/env python3
"""Summarise the gfx950 ISA of the vmk kernels from a hipcc --save-temps .s file: registers, scratch, instruction mix.
usage: tools/isa_summary.py file.s [name-substring ...]"""
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2:] or ["k_render", "k_trace", "k_test", "k_aov", "_ool"]
    text = open(path).read().split("\n")
    # function bodies
    funcs = {}
    cur = None
    for ln in text:
        m = re.match(r"^(_Z[\w]+):\s*(;.*)?$", ln)
        if m:
            cur = m.group(1); funcs[cur] = []
            continue
        if ln.startswith(".Lfunc_end") or ln.startswith("\t.section"):
            cur = None
        if cur and ln.startswith("\t") and not ln.startswith("\t."):
            funcs[cur].append(ln.strip())
    meta = {}
    name = None
    for ln in text:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1); meta[name] = {}
        m = re.match(r"\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):\s+(\d+)", ln)
        if m and name:
            meta[name][m.group(1)] = int(m.group(2))
    for f, body in funcs.items():
        if not any(w in f for w in want):
            continue
        ops = [b.split()[0] for b in body if b and not b.startswith(";")]
        cnt = lambda p: sum(1 for o in ops if o.startswith(p))
        md = meta.get(f, {})
        print(f"{f[:70]:70s} n={len(ops):6d} valu={cnt('v_'):6d} salu={cnt('s_'):6d} ds={cnt('ds_'):4d} global={cnt('global_'):4d} flat={cnt('flat_'):4d} "
              f"scratch_ld={cnt('scratch_load'):4d} scratch_st={cnt('scratch_store'):4d} buffer={cnt('buffer_'):4d} swappc={cnt('s_swappc'):3d} | "
              f"vgpr={md.get('vgpr_count', '-')} sgpr={md.get('sgpr_count', '-')} scratch={md.get('private_segment_fixed_size', '-')} vspill={md.get('vgpr_spill_count', '-')} lds={md.get('group_segment_fixed_size', '-')}")


if __name__ == "__main__":
    main()
