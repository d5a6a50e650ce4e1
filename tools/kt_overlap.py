#!/usr/bin/env python3
"""tools/kt_overlap.py PREFIX -- the last render call of a rocprofv3 --kernel-trace csv: per stream the span of its kernels, the sum of their
durations and the time at least one / at least two kernels were running."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "_kernel_trace.csv")))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"]) for r in rows)
gens = [i for i, k in enumerate(ks) if "k_generate" in k[3]]
n_l = len(set(k[2] for k in ks if "k_generate" in k[3]))
i0 = gens[-n_l]
call = ks[i0:]
t0 = call[0][0]
by = {}
for a, b, s, n in call:
    by.setdefault(s, []).append((a - t0, b - t0))
for s, v in sorted(by.items()):
    print("stream %s: %3d kernels, first start %.3f ms, last end %.3f ms, sum of durations %.3f ms" % (s, len(v), v[0][0] / 1e6, max(b for a, b in v) / 1e6, sum(b - a for a, b in v) / 1e6))
ev = sorted([(a, 1) for a, b, s, n in call] + [(b, -1) for a, b, s, n in call])
lvl = 0; last = ev[0][0]; t = [0] * 16
for x, d in ev:
    t[min(lvl, 15)] += x - last; last = x; lvl += d
print("time with k kernels running (ms):", {k: round(v / 1e6, 3) for k, v in enumerate(t) if v})
