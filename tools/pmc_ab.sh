#!/bin/bash
# tools/pmc_ab.sh TAG WORKLOAD "ENV1" "ENV2" ... -- SQ instruction counters of the traversal kernels for several option settings
TAG=$1; W=$2; shift; shift
export TMPDIR=/tmp
k=0
for E in "$@"; do
  k=$((k+1))
  env $E true
  ( export $E; timeout -k 10 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmcab_${TAG}_$k -- python3 bench.py --workload $W --steps 1 --profile > gpurun_out/pmcab_${TAG}_$k.log 2>&1 )
  python3 tools/summarize_pmc.py gpurun_out/pmcab_${TAG}_$k.csv gpurun_out/pmcab_${TAG}_$k
  rm -rf gpurun_out/pmcab_${TAG}_$k
  echo "## $E"; grep "k_extend\|k_connect" gpurun_out/pmcab_${TAG}_$k.csv | awk -F, '{n=split($0,a,","); printf "  %-22s %-34s %14.0f\n", a[1], a[2]a[3]a[4]a[5]a[6], a[n]}'
done
