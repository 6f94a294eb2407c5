#!/bin/bash
OUT=gpurun_out/r4c
mkdir -p $OUT
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do
cp ab_libs/old.so modular-slam_amd/libmslam_hip.so
python tools/stage_times.py --reps 8 --label "old" 2>&1 | grep "^\[.*median"
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
for G in 1 2 3 4 6 8 16; do
  MSLAM_HIP_FAST_GROUP=$G python tools/stage_times.py --reps 8 --label "g$G" 2>&1 | grep "^\[.*median"
done
done
