#!/bin/bash
# A/B of tdx_tune_set knobs on the training leg, ON the GPU box:  bash tools/sweep_tune.sh "knob=v,knob=v" "..." ...
# Each argument is one TDX_TUNE setting ("" = defaults); prints ms/step of `bench.py --train-only` for each.
for t in "$@"; do
  echo -n "[$t] "
  TDX_TUNE=$t python bench.py --train-only --steps ${STEPS:-40} --warmup 10 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done
