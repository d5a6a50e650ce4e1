#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's -save-temps assembly.

    python tools/isa_mix.py FILE.s SUBSTRING_OF_MANGLED_NAME

Counts by class (f32 VALU, f64 VALU, transcendental, div helpers, memory, LDS, SALU, branches) and prints the
register / scratch / occupancy lines of the kernel's metadata.  Static counts: every divergent path counts once.
"""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][^\s]*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    body = []
    for l in lines[start + 1:]:
        if l.startswith("\t.section") or l.startswith(".Lfunc_end"):
            break
        body.append(l)
    c = Counter()
    ops = Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        ops[op] += 1
        if op.startswith("v_div_") or op.startswith("v_rcp") or op.startswith("v_rsq"):
            c["div/rcp helpers"] += 1
        elif op.startswith("v_sqrt"):
            c["sqrt"] += 1
        elif op.startswith("v_") and "f64" in op:
            c["valu f64"] += 1
        elif op.startswith("v_cmp") or op.startswith("v_cndmask"):
            c["valu cmp/select"] += 1
        elif op.startswith("v_"):
            c["valu other"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            c["vmem " + ("load" if "load" in op else "store/atomic")] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            c["branch"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        else:
            c["other"] += 1
    tot = sum(c.values())
    print("kernel:", lines[start][:100])
    print("static instructions:", tot)
    for k, v in c.most_common():
        print("  %-20s %6d  %5.1f %%" % (k, v, 100.0 * v / tot))
    print("top opcodes:", ", ".join("%s %d" % kv for kv in ops.most_common(25)))
    for l in lines[start:]:
        if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumVgprs|NumSgprs)", l):
            print(l.strip())
        if l.startswith("\t.section") and l is not lines[start]:
            pass
        if "; Occupancy" in l:
            break


if __name__ == "__main__":
    main()
