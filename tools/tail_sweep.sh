#!/bin/bash
# tools/tail_sweep.sh -- the hand-over threshold of the fused tail (paths per segment): fixed-cost probe for each value
for T in ${TS:-32 64 96 192 384 768 1536}; do
  echo "## tail_paths $T"
  PTRS_OPT_TAIL_PATHS=$T python tools/fixed_cost_probe.py
done
