#!/bin/bash
# tools/shade_halves.sh WORKLOAD -- the NEE / continuation split of the shade kernels, measured with ablation builds (tools/ablate.sh h1 "-DPTRS_ABL_SHADE_HALF=1":
# vertices without their continuation; h2 "-DPTRS_ABL_SHADE_HALF=2": without their next-event estimation; timing only): duration of the FIRST launch of every shade
# kernel (round 0: the same vertices in all three builds) from a rocprofv3 kernel trace of a single-lane frame
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
W=${1:-cornell}
for V in hip h1 h2; do
  export PTRS_LIB=pathtracer-rs_amd/libptrs_$V.so
  rm -rf gpurun_out/sh_$V
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sh_$V -- python3 bench.py --workload $W --profile --steps 1 --tail-at -1 > gpurun_out/sh_$V.log 2>&1
  F=$(find gpurun_out/sh_$V -name "*kernel_trace.csv" | head -1)
  echo "## $W $V"; python3 - "$F" <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seen = {}
tot = {}
for r in rows:
    m = re.search(r"(k_shade<[^>(]*>)", r["Kernel_Name"])
    if not m: continue
    n = m.group(1); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[n] = tot.get(n, 0.0) + d
    if n not in seen: seen[n] = d
for n in seen: print("  %-24s first launch (round 0) %8.3f ms   all launches %8.2f ms" % (n, seen[n], tot[n]))
PY
  rm -rf gpurun_out/sh_$V
done
