#!/usr/bin/env python3
"""tools/kres.py [extra hipcc flags] -- registers, scratch, LDS and occupancy of every kernel of libptrs_hip (hipcc's kernel-resource-usage remarks)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
b = importlib.import_module("pathtracer-rs_amd.build")
cmd = [b.HIPCC] + b.FLAGS + sys.argv[1:] + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kres.so", os.path.join(b.CSRC, "ptrs_hip.hip")]
out = open(sys.argv.pop(sys.argv.index("--from") + 1)).read() if "--from" in sys.argv else subprocess.run(cmd, capture_output=True, text=True).stderr
cur, d = None, {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); d[cur] = {}
    for k, short in (("VGPRs:", "v"), ("AGPRs:", "a"), ("TotalSGPRs:", "s"), ("ScratchSize [bytes/lane]:", "scr"), ("Occupancy [waves/SIMD]:", "occ"), ("LDS Size [bytes/block]:", "lds"), ("VGPRs Spill:", "vspill"), ("SGPRs Spill:", "sspill")):
        m = re.search(r"remark:\s+" + re.escape(k) + r" (\d+)", l)
        if m and cur:
            d[cur][short] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
for k, n in sorted(zip(d, names), key=lambda t: t[1]):
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    print("%-64s %s" % (n[:64], " ".join("%s=%d" % kv for kv in d[k].items())))
