#!/usr/bin/env python3
"""tools/call_timeline.py KERNEL_TRACE.csv [CALL] -- the kernels of the CALL-th render call of a traced run (calls are separated by the
k_generate launches that follow a gap), in start order: offset from the call's first kernel, duration, queue, name; and the time with
no kernel in flight."""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
call = int(sys.argv[2]) if len(sys.argv) > 2 else -1
# split into calls: a gap of > 0.5 ms with nothing in flight
calls, cur, end = [], [], 0
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur and a - end > 500000:
        calls.append(cur); cur = []
    cur.append(r); end = max(end, b)
calls.append(cur)
c = calls[call]
t0 = int(c[0]["Start_Timestamp"])
idle, end = 0, t0
for r in c:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a > end:
        idle += a - end
    end = max(end, b)
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:48]
    print("%9.3f ms  +%8.3f ms  q%-3s %s" % ((a - t0) / 1e6, (b - a) / 1e6, r.get("Queue_Id", "?"), name))
print("calls %d; this call: %d kernels, wall %.3f ms, nothing in flight for %.3f ms" % (len(calls), len(c), (end - t0) / 1e6, idle / 1e6))
