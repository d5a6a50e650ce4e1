#!/bin/bash
# tools/opt_sweep.sh WORKLOAD "ENV1" "ENV2" ... -- bench.py once per option setting (PTRS_OPT_* of the Python host): ms per step, one line each
W=$1; shift
for E in "$@"; do
  env $E python bench.py --workload $W --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-collective-smoke 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('$E', round(j['ms_per_step'], 2), {k[:7]: round(v, 1) for k, v in r['single_lane_frame_ms'].items()}, j['film_check'])
"
done
