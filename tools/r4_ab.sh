#!/bin/bash
# parity on ab_libs/new.so, then the A/B against ab_libs/old.so (stage times + bench)
OUT=gpurun_out/${1:-r4ab}
mkdir -p $OUT
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/ab_libs.sh > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
