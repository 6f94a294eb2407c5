#!/bin/bash
# step-level A/B: ab_libs/{old,new}.so alternated N times, 60-step bench runs
N=${1:-4}
for i in $(seq $N); do
for v in old new; do
  cp ab_libs/$v.so modular-slam_amd/libmslam_hip.so
  python bench.py --no-cpu-baseline --no-extras --steps 60 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] %.1f M kp/s, %.3f ms/step' % (d['value']/1e6, d['ms_per_step']), {k: round(v,3) for k,v in d['roofline']['stages_ms_per_step'].items()})"
done
done
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
