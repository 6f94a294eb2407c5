#!/bin/bash
# env-knob sweep on the in-tree library: stage_times per setting ("" = default), two rounds
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python tools/stage_times.py --reps 8 --label "$cfg" 2>&1 | grep "^\[.*median"
done
done
