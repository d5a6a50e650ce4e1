#!/bin/bash
# tools/prof.sh WORKLOAD TAG [FRAMES] -- one rocprofv3 kernel-trace pass and the PMC passes of `bench.py --profile`
# (single pipeline lane, FRAMES frames).  Counters are collected in their own runs (no trace domains beside --pmc).
# Output: gpurun_out/prof_TAG/{kernel_stats.csv,pmc.csv,pmc.json,report.txt}; pmc.json carries the hash of the kernel sources
# (pathtracer-rs_amd/build.py source_hash) it was measured on, which bench.py checks before it uses the file.
set -e
W=${1:-cornell}; TAG=${2:-x}; FR=${3:-1}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python3 bench.py --workload $W --steps $FR --profile"
case $W in trace-*) ;; *) # the profiled single-lane frame hands over to the fused tail where the bench's own single-lane frame does (an unprofiled run finds the round)
  K=$(python3 bench.py --workload $W --print-tail-at 2> "$OUT/tail_at.log" | tail -n 1); echo "tail_at $K"; CMD="$CMD --tail-at $K";;
esac
case $W in trace-*) # the ray sets are made by an unprofiled run, so that the profiled processes launch the traced kernel only
  rm -f /tmp/rays_$W.npz; python3 bench.py --workload $W --rays /tmp/rays_$W.npz > "$OUT/rays.log" 2>&1
  CMD="$CMD --rays /tmp/rays_$W.npz";;
esac
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $CMD > "$OUT/kt.log" 2>&1
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_MUL_F64"; do
  i=$((i+1))
  echo "pmc pass $i: $SET"
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed (see $OUT/pmc$i.log)"
done
python3 tools/summarize_pmc.py "$OUT/pmc.csv" "$OUT"/pmc[0-9]
python3 tools/prof_report.py "$OUT" --json "$OUT/pmc.json" --frames $FR --workload $W > "$OUT/report.txt"
rm -rf "$OUT"/kt "$OUT"/pmc[0-9]
echo "done: $OUT"
