#!/bin/bash
# tools/trace_profiles.sh -- the traversal-bench part of tools/final_profiles.sh alone
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
set -x
OUT=gpurun_out/final; mkdir -p $OUT
for W in trace-colonnade trace-classroom; do
  timeout -k 10 900 tools/prof.sh $W fin_$W 3 || exit 1
  cp gpurun_out/prof_fin_$W/pmc.json profiles/r04_pmc_$W.json
  cp gpurun_out/prof_fin_$W/pmc.json $OUT/r04_pmc_$W.json
  cp gpurun_out/prof_fin_$W/pmc.csv $OUT/r04_${W}_pmc.csv
  cp gpurun_out/prof_fin_$W/kernel_stats.csv $OUT/r04_${W}_kernel_stats.csv
  cp gpurun_out/prof_fin_$W/report.txt $OUT/r04_${W}_report.txt
done
timeout -k 10 600 python bench.py --workload trace-colonnade --steps 5 > $OUT/r04_bench_trace-colonnade.json || exit 1
timeout -k 10 600 python bench.py --workload trace-classroom --steps 5 > $OUT/r04_bench_trace-classroom.json || exit 1
timeout -k 10 600 python bench.py --workload trace-colonnade --steps 5 --node-order 1 > $OUT/r04_bench_trace-colonnade_treelets.json || exit 1
