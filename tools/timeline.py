#!/usr/bin/env python3
"""tools/timeline.py KERNEL_TRACE.csv -- how much of the wall time of a traced render had 1, 2, 3 ... kernels in flight, and per stream
(queue) the busy time; from rocprofv3 --kernel-trace output."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per_q = defaultdict(float)
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    per_q[q] += (b - a) / 1e6
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth, last, hist = 0, t0, defaultdict(float)
for t, d in ev:
    hist[depth] += (t - last) / 1e6
    depth += d; last = t
print("wall %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
for k in sorted(hist):
    print("  %d kernels in flight: %8.1f ms" % (k, hist[k]))
for q, v in sorted(per_q.items()):
    print("  queue %s busy %.1f ms" % (q, v))
