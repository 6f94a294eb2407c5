#!/bin/bash
# round 6: parity + A/B of the one-launch level chain (k_level_chain) against the per-level launches
TAG=${1:-r6b}
mkdir -p gpurun_out/$TAG
MSLAM_HIP_LEVEL_CHAIN=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$TAG/pytest_chain.log 2>&1
echo "pytest (chain on) rc=$?"; tail -3 gpurun_out/$TAG/pytest_chain.log
for cfg in "" "MSLAM_HIP_LEVEL_CHAIN=1" "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_COH=0" \
           "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_WAVES=8" "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_WAVES=2" \
           "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_FRAMES=2 MSLAM_HIP_LEVEL_CHAIN_WAVES=8" \
           "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_FRAMES=2" "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_FRAMES=4 MSLAM_HIP_LEVEL_CHAIN_WAVES=8" \
           "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_K6=4" "MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_FRAMES=8 MSLAM_HIP_LEVEL_CHAIN_WAVES=8"; do
  env $cfg timeout -k 10 120 python tools/stage_times.py --reps 8 --label "$cfg" 2>/dev/null | grep "median" >> gpurun_out/$TAG/stage_ab.txt || echo "FAILED $cfg"
done
cat gpurun_out/$TAG/stage_ab.txt
