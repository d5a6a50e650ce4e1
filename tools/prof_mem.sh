#!/bin/bash
# tools/prof_mem.sh WORKLOAD TAG -- vector-memory pipeline counters (TA / TCP / TD / UTCL1) of `bench.py --profile`
set -e
W=${1:-cornell}; TAG=${2:-x}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/mem_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --workload $W --steps 1 --profile ${EXTRA}"
i=0
for SET in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  echo "pmc pass $i: $SET"
  timeout -k 10 150 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/pmc$i.log"; }
done
python3 tools/summarize_pmc.py "$OUT/pmc.csv" "$OUT"/pmc[0-9]*/
rm -rf "$OUT"/pmc[0-9]*/
grep -v "rocclr\|at::native\|reduce_counts" "$OUT/pmc.csv" | grep "k_shade\|k_extend\|k_connect\|counter"
