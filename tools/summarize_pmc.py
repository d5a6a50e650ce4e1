#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (counter, kernel).

    python tools/summarize_pmc.py OUT.csv DIR [DIR ...]

Each DIR is the -d directory of one `rocprofv3 --pmc <COUNTER> --output-format csv` pass.
Kernel names are shortened to the identifier before '<' / '('.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", name)  # keeps the template arguments: k_shade<0, 0> and k_shade<4, 3> are different kernels
    return m.group(1).replace(" ", "") if m else name[:40]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: [0, 0.0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = (row["Counter_Name"], short(row["Kernel_Name"]))
                    acc[k][0] += 1
                    acc[k][1] += float(row["Counter_Value"])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["counter", "kernel", "launches", "sum", "per_launch"])
        for (c, k), (n, s) in sorted(acc.items()):
            w.writerow([c, k, n, s, s / max(n, 1)])


if __name__ == "__main__":
    main()
