#!/bin/bash
# round 6: SQ counters of the one-launch level chain (k_level_chain, 2 frames / 8 waves per workgroup) next to the per-level
# launches, from the serialized stage pass (one 1000-frame launch each); plus a TA busy / GRBM_GUI_ACTIVE pass (calibration of
# the TA busy fraction bench.py reports).  The program follows `--` directly; the switches are exported, not passed through env.
TAG=${1:-r6e}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY"
for arm in off on; do
  if [ $arm = on ]; then export MSLAM_HIP_LEVEL_CHAIN=1 MSLAM_HIP_LEVEL_CHAIN_FRAMES=2 MSLAM_HIP_LEVEL_CHAIN_WAVES=8; fi
  timeout -k 10 200 rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_$arm" -o sq -- python3 tools/stage_times.py --reps 2 > "$OUT/sq_$arm.out" 2> "$OUT/sq_$arm.err" || { echo "SQ pass $arm failed"; tail -3 "$OUT/sq_$arm.err"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta_$arm" -o ta -- python3 tools/stage_times.py --reps 2 > "$OUT/ta_$arm.out" 2> "$OUT/ta_$arm.err" || { echo "TA pass $arm failed"; tail -3 "$OUT/ta_$arm.err"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$arm" -o kt -- python3 tools/stage_times.py --reps 6 > "$OUT/kt_$arm.out" 2> "$OUT/kt_$arm.err" || { echo "kt pass $arm failed"; exit 1; }
  python tools/summarize_sq.py "$OUT/sq_$arm/sq_counter_collection.csv" > "$OUT/sq_$arm.txt" 2>&1
  python tools/summarize_sq.py "$OUT/ta_$arm/ta_counter_collection.csv" > "$OUT/ta_$arm.txt" 2>&1
done
grep -A9 "k_level_chain\|k_gray_blur\|k_resize_blur\|k_describe\|k_fast_cells" "$OUT"/sq_*.txt "$OUT"/ta_*.txt | cut -c1-200
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +4M -delete
head -30 "$OUT"/kt_on/kt_kernel_stats.csv | cut -c1-220
