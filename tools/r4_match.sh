#!/bin/bash
OUT=gpurun_out/${1:-r4i}
mkdir -p $OUT
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_pnp.py -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do
for v in old new; do
  cp ab_libs/$v.so modular-slam_amd/libmslam_hip.so
  python tools/stage_times.py --reps 8 --label "$v cfg2" 2>&1 | grep "^\[.*median"
  python tools/stage_times.py --reps 8 --label "$v cfg5" --width 1920 --height 1080 --levels 3 --batch 64 --min-area 370 2>&1 | grep "^\[.*median"
done
done
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
