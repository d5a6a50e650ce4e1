#!/bin/bash
# tools/ab_opt.sh TAG "ENV_A" "ENV_B" [ROUNDS] -- ABAB of two option settings (PTRS_OPT_* of the Python host) on the three frame workloads
TAG=$1; A=$2; B=$3; N=${4:-2}
OUT=gpurun_out/abopt_$TAG.txt
: > $OUT
for i in $(seq $N); do
  for E in "$A" "$B"; do
    for W in ${WORKLOADS:-cornell colonnade classroom}; do
      echo "## $E $W" >> $OUT
      env $E python bench.py --workload $W --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-collective-smoke 2>> gpurun_out/abopt_$TAG.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(json.dumps({'value': round(j['value'], 1), 'ms_per_step': round(j['ms_per_step'], 2), 'single_lane_ms': {k[:7]: round(v, 1) for k, v in r['single_lane_frame_ms'].items()}, 'tail': j['config']['launch']['fused_tail'], 'launches': j['config']['launch']['kernel_launches_per_frame'], 'film_check': j['film_check']}))
" >> $OUT
    done
  done
done
cat $OUT
