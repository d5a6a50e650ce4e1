#!/bin/bash
# tools/ab_lib.sh TAG LIB_A LIB_B [ROUNDS] -- ABAB of two builds of the library (PTRS_LIB) on the three frame workloads: ms per step and the
# single-lane kernel-class times of every run into gpurun_out/ablib_TAG.txt
TAG=$1; A=$2; B=$3; N=${4:-2}
OUT=gpurun_out/ablib_$TAG.txt
: > $OUT
for i in $(seq $N); do
  for L in $A $B; do
    for W in ${WORKLOADS:-cornell colonnade classroom}; do
      echo "## $L $W" >> $OUT
      PTRS_LIB=$L python bench.py --workload $W --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-collective-smoke 2>> gpurun_out/ablib_$TAG.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(json.dumps({'value': round(j['value'], 1), 'ms_per_step': round(j['ms_per_step'], 2), 'single_lane_ms': {k[:7]: round(v, 1) for k, v in r['single_lane_frame_ms'].items()}, 'film_check': j['film_check']}))
" >> $OUT
    done
  done
done
cat $OUT
