#!/bin/bash
TAG=${1:-r4d}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
bash tools/pmc_quick.sh $TAG
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq3" -o sq -- python3 tools/stage_times.py --reps 2 > "$OUT/sq3.out" 2> "$OUT/sq3.err"
python tools/summarize_sq.py "$OUT/sq3/sq_counter_collection.csv" "$OUT/pmc_sq3.json" > "$OUT/pmc_sq3.txt" 2>&1
cat "$OUT/pmc_sq3.txt" | cut -c1-400
tail -3 "$OUT/sq3.err"
