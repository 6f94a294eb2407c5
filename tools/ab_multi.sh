#!/bin/bash
# A/B of several builds of libmslam_hip.so in ONE gpurun call: every ab_libs/<name>.so given as argument is copied over the
# in-tree library in turn (two rounds), "name:VAR=1" adds an environment setting.  usage (through gpurun): bash tools/ab_multi.sh old new "new:MSLAM_X=0"
cp modular-slam_amd/libmslam_hip.so /tmp/keep.so
for rep in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; e=""; [ "$v" != "$spec" ] && e=${spec#*:}
    cp ab_libs/$v.so modular-slam_amd/libmslam_hip.so
    env $e python tools/stage_times.py --reps 8 --label "$spec" 2>&1 | grep "^\[.*median"
  done
done
for spec in "$@" "$@"; do
  v=${spec%%:*}; e=""; [ "$v" != "$spec" ] && e=${spec#*:}
  cp ab_libs/$v.so modular-slam_amd/libmslam_hip.so
  env $e python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$spec] %.1f M kp/s, %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"
done
cp /tmp/keep.so modular-slam_amd/libmslam_hip.so
