#!/bin/bash
# tools/clock_watch.sh WORKLOAD -- samples rocm-smi clocks / power / temperature every 0.25 s while bench.py runs (diagnostic)
W=${1:-cornell}
OUT=gpurun_out/clock_$W.txt
: > $OUT
python bench.py --workload $W --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/clock_bench_$W.json 2> gpurun_out/clock_bench_$W.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)|GPU use" | tr '\n' ';' >> $OUT
  echo >> $OUT
  sleep 0.25
done
wait $BP
echo "samples: $(wc -l < $OUT)"
