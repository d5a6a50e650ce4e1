#!/bin/bash
# tools/ab.sh TAG "ENV1" "ENV2" ... -- bench.py on the three workloads for each environment setting (PTRS_OPT_* knobs),
# JSON lines into gpurun_out/ab_TAG.txt
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
for E in "$@"; do
  for W in ${WORKLOADS:-cornell colonnade classroom}; do
    echo "## $E $W" >> $OUT
    env $E python bench.py --workload $W --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-collective-smoke 2>> gpurun_out/ab_$TAG.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(json.dumps({'value': round(j['value'], 1), 'ms_per_step': round(j['ms_per_step'], 2), 'single_lane_ms': {k: round(v, 1) for k, v in r['single_lane_frame_ms'].items()}, 'film_check': j['film_check']}))
" >> $OUT
  done
done
cat $OUT
