#!/bin/bash
# tools/final_profiles.sh -- everything under profiles/r04_* from ONE tree on one GPU box: the issue-rate table, the counter sets of the
# three frame workloads and the two traversal workloads, and the bench lines that read them.  Writes into gpurun_out/final/; copy to
# profiles/ afterwards (tools/collect_profiles.py).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -x
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
# (the issue-rate table profiles/r03_valu_ceiling.json is a property of the hardware, not of this tree: tools/bin/valu_ceiling re-measures it)
for W in cornell colonnade classroom; do
  FR=1
  timeout -k 10 900 tools/prof.sh $W fin_$W $FR || exit 1
  cp gpurun_out/prof_fin_$W/pmc.json profiles/r04_pmc_$W.json
  cp gpurun_out/prof_fin_$W/pmc.json $OUT/r04_pmc_$W.json
  cp gpurun_out/prof_fin_$W/pmc.csv $OUT/r04_${W}_pmc.csv
  cp gpurun_out/prof_fin_$W/kernel_stats.csv $OUT/r04_${W}_kernel_stats.csv
  cp gpurun_out/prof_fin_$W/report.txt $OUT/r04_${W}_report.txt
done
timeout -k 10 600 python bench.py > $OUT/r04_bench_cornell.json || exit 1
timeout -k 10 600 python bench.py --workload colonnade --no-collective-smoke > $OUT/r04_bench_colonnade.json || exit 1
timeout -k 10 600 python bench.py --workload classroom --no-collective-smoke > $OUT/r04_bench_classroom.json || exit 1
if [ -z "$SKIP_TRACE" ]; then tools/trace_profiles.sh || exit 1; fi
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/r04_bench_*.json")):
    j = json.load(open(f)); r = j["roofline"]
    print(f, "%.0f %s, %.1f ms, bound %s, frac %s" % (j["value"], j["unit"], j["ms_per_step"], r.get("bound"), r.get("frac")))
PY
