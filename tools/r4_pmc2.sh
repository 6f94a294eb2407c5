#!/bin/bash
# memory-path counters per kernel of the serialized stage pass (what holds resize / describe below both roofs?)
# SQ groups only: a TA_* / TCP_* group left "incomplete dispatches" behind and ran into rocprofv3's five-minute limit twice
# on this pool (round 4: 10 GPU-minutes for nothing) — do not add them back.
TAG=${1:-r4p}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
i=0
for G in "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G --output-format csv -d "$OUT/g$i" -o g -- python3 tools/stage_times.py --reps 2 > "$OUT/g$i.out" 2> "$OUT/g$i.err"
  python tools/summarize_sq.py "$OUT/g$i/g_counter_collection.csv" "$OUT/g$i.json" > "$OUT/g$i.txt" 2>&1
  tail -2 "$OUT/g$i.err" | cut -c1-200
done
for k in k_resize_blur k_gray_blur k_describe k_fast_cells; do for i in 1 2; do echo "== $k g$i"; grep -A9 "$k" "$OUT/g$i.txt" | head -10; done; done
