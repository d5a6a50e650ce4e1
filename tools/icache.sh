#!/bin/bash
# tools/icache.sh -- instruction-cache counters of the Cornell frame, one pipeline lane vs the default four (kernels of different
# passes share the CUs: 17 + 17 + 50 KB of code against a 64 KB instruction cache per CU pair)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
for L in 1 0; do
  rm -rf gpurun_out/ic$L
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ic$L -- python3 bench.py --profile --steps 2 --profile-lanes $L > gpurun_out/ic$L.log 2>&1
  python3 tools/summarize_pmc.py gpurun_out/ic$L.csv gpurun_out/ic$L
  rm -rf gpurun_out/ic$L
  echo "## lanes $L"; python3 - <<PY
import csv, collections
t = collections.defaultdict(float); k = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open("gpurun_out/ic$L.csv")):
    t[r["counter"]] += float(r["sum"]); k[r["kernel"]][r["counter"]] += float(r["sum"])
print({c: round(v) for c, v in t.items()})
print("hit rate %.4f, misses per 1000 fetches %.2f" % (t["SQC_ICACHE_HITS"] / max(t["SQC_ICACHE_REQ"], 1), 1000 * t["SQC_ICACHE_MISSES"] / max(t["SQC_ICACHE_REQ"], 1)))
for n, c in k.items():
    if c.get("SQC_ICACHE_REQ", 0) > 1e6: print("  %-36s req %.3g miss %.3g (%.3f)" % (n, c["SQC_ICACHE_REQ"], c["SQC_ICACHE_MISSES"], c["SQC_ICACHE_MISSES"] / c["SQC_ICACHE_REQ"]))
PY
done
