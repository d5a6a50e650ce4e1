#!/usr/bin/env python3
"""tools/collect_profiles.py -- copies what tools/final_profiles.sh left in gpurun_out/final/ into profiles/ (tracked)."""
import glob
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 0
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "final", "r04_*"))):
    shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
    n += 1
print("copied %d files" % n)
