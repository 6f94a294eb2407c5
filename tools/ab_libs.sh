#!/bin/bash
# A/B of two builds of libmslam_hip.so in ONE gpurun call (boxes differ by up to 10 %): ab_libs/{old,new}.so are copied over
# the in-tree library in turn.  usage (through gpurun): bash tools/ab_libs.sh [stage_times args]
for rep in 1 2; do
  for v in old new; do
    cp ab_libs/$v.so modular-slam_amd/libmslam_hip.so
    python tools/stage_times.py --reps 8 --label "$v" "$@" 2>&1 | grep "^\[.*median"
  done
done
for v in old new old new; do
  cp ab_libs/$v.so modular-slam_amd/libmslam_hip.so
  python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] %.1f M kp/s, %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
done
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
