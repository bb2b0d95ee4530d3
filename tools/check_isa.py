#!/usr/bin/env python3
"""What the compiler made of the kernels: per instantiation of sw_pipe_kernel / sw_lane_kernel / sw_sp_kernel, the VGPRs, the
scratch bytes, and -- for the pipeline kernel's binary16 tier -- the `v_perm_b32` and `v_lshrrev_b32 v, 16, v` counts.

    make -C swimm_amd/csrc isa && python tools/check_isa.py [swimm_amd/csrc/obj/sw_kernels.s]

The column loop of the binary16 tier needs exactly 4 columns x T `v_pk_fma_f16` (a pair's two profile dwords + the diagonal,
pair_score_plus), no `v_perm_b32` and no 16-bit shift at all (the integer tiers: 4 x T `v_perm_b32`).  Anything beyond that is a
re-pack the compiler added: round 3 found T shifts + T perms at the top of EVERY step in the 20-, 24- and 28-row group-resident
instantiations (AMDGPUPromoteAllocaToVector had made the E[] register array one <2T x half> vector), 5-8 % of their
throughput.  tests/test_codegen.py keeps that from coming back unnoticed.
"""
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def kernels(path):
    txt = open(path).read()
    out = []
    for m in re.finditer(r"\n(_ZN5swimm\d+((?:sw_|retile|tile_sequences|publish_items)\w*?_kernel)(\w*)):[^\n]*\n(.*?)s_endpgm(.*?)\.end_amdhsa_kernel", txt, re.S):
        sym, kind, rest, body, meta = m.groups()
        rec = {"symbol": sym, "kernel": kind,
               "vgprs": int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)),
               "scratch_bytes": int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)),
               "instructions": sum(1 for l in body.split("\n") if l.startswith("\t") and not l.lstrip().startswith((";", ".")))}
        t = re.match(r"ILi(\d+)ELi(\d)ELb(\d)ELb(\d)ELb(\d)E", rest)
        if kind == "sw_pipe_kernel" and t:
            rec.update(rows_per_wave=int(t.group(1)), mode=int(t.group(2)), dynamic=t.group(3) == "1", group_resident=t.group(4) == "1", growing_list=t.group(5) == "1",
                       v_perm_b32=len(re.findall(r"\bv_perm_b32", body)), v_pk_fma_f16=len(re.findall(r"\bv_pk_fma_f16", body)),
                       shifts_by_16=len(re.findall(r"v_lshrrev_b32_e32 v\d+, 16, v\d+", body)))
        t = re.match(r"ILi(\d)ELi(\d+)E", rest)
        if kind == "sw_lane_kernel" and t:
            rec.update(mode=int(t.group(1)), rows_per_lane=int(t.group(2)))
        out.append(rec)
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "swimm_amd", "csrc", "obj", "sw_kernels.s")
    for k in kernels(path):
        print(json.dumps(k))


if __name__ == "__main__":
    main()
