#!/usr/bin/env python3
"""tools/isa_classes.py FILE.s KERNEL_SUBSTRING [--blocks] -- issue classes of a kernel's instructions on gfx950.

Class A = the vector instructions tools/valu_ceiling.hip measures at the full rate (fp32 add / sub / mul / fma / mac, and / or / xor,
u32 add / sub, mov); class B = every other vector ALU instruction (one port, ~4.2 cycles: comparisons, v_cndmask, min / max,
shifts, integer mads, conversions, fp64, packed fp32, transcendental); then SALU, LDS, VMEM, branches.  With --blocks the
counts are listed per basic block (loops show as the blocks between their header label and the back branch).
A time estimate per block: 4.2 x max(B, (A + B) / 2) cycles per wave-instruction stream (needs >= 2 waves per SIMD)."""
import re
import sys
from collections import Counter

A_OPS = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
         "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32")


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_"):
        return "A" if base in A_OPS else "B"
    if base.startswith("ds_"):
        return "LDS"
    if base.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    if base.startswith(("s_cbranch", "s_branch")):
        return "BR"
    if base.startswith("s_waitcnt") or base.startswith("s_nop"):
        return "WAIT"
    if base.startswith("s_"):
        return "SALU"
    return "?"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][^\s]*:", l) and key in l.split(":")[0]), None)
    if start is None:
        sys.exit("kernel not found")
    blocks, cur, name = [], Counter(), "entry"
    ops = Counter()
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if re.match(r"^\.LBB[0-9_]+:", t):
            blocks.append((name, cur)); cur = Counter(); name = t.split(":")[0] + " " + (t.split(";")[1].strip() if ";" in t else "")
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        c = classify(op)
        cur[c] += 1
        if c == "B":
            ops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)] += 1
    blocks.append((name, cur))
    tot = Counter()
    for _, c in blocks:
        tot.update(c)
    est = lambda c: 4.2 * max(c["B"], (c["A"] + c["B"]) / 2.0)
    if "--blocks" in sys.argv:
        for n, c in blocks:
            if sum(c.values()) >= 8:
                print("%-70s A=%3d B=%3d SALU=%3d LDS=%2d VMEM=%2d BR=%2d  ~%4.0f cyc" % (n[:70], c["A"], c["B"], c["SALU"], c["LDS"], c["VMEM"], c["BR"], est(c)))
    print("total: A=%d B=%d SALU=%d LDS=%d VMEM=%d BR=%d" % (tot["A"], tot["B"], tot["SALU"], tot["LDS"], tot["VMEM"], tot["BR"]))
    print("class B by opcode:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
