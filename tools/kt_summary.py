#!/usr/bin/env python3
"""tools/kt_summary.py PREFIX -- kernels of a rocprofv3 --kernel-trace csv (PREFIX_kernel_trace.csv) by name and grid: calls, average and total time."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1] + "_kernel_trace.csv")))
acc = defaultdict(lambda: [0, 0])
for r in rows:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"]).split("::")[-1]
    nm = re.sub(r"^void ", "", nm)
    k = (nm[:44], r["Grid_Size_X"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"])
    acc[k][0] += 1; acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-44s grid %8s lds %6s scratch %5s vgpr %4s calls %4d avg %8.1f us total %8.2f ms" % (k + (n, t / n / 1e3, t / 1e6)))
