#!/bin/bash
# round 5, first GPU call: counter list of this box, baseline stage times, one SQ pass on the LDS side of the serialized stages
TAG=${1:-r5a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
timeout -k 10 120 rocprofv3 -L > "$OUT/counters_avail.txt" 2> "$OUT/counters_avail.err"
wc -l "$OUT/counters_avail.txt"
timeout -k 10 200 python tools/stage_times.py --reps 8 --label base > "$OUT/stage_times.txt" 2>&1 && grep median "$OUT/stage_times.txt"
timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/lds" -o g -- python3 tools/stage_times.py --reps 2 > "$OUT/lds.out" 2> "$OUT/lds.err"
echo "lds pass rc=$?"; tail -2 "$OUT/lds.err" | cut -c1-300
python tools/summarize_sq.py "$OUT/lds/g_counter_collection.csv" "$OUT/lds.json" > "$OUT/lds.txt" 2>&1
for k in k_describe k_fast_cells k_quadtree; do echo "== $k"; grep -A9 "$k" "$OUT/lds.txt" | head -10; done
find "$OUT" -name "*.csv" -size +8M -delete
