#!/bin/bash
# quick per-kernel SQ counters of the cfg2 step (2 repetitions of the serialized stage pass); run through gpurun
TAG=${1:-pmcq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -o sq -- python3 tools/stage_times.py --reps 2 > "$OUT/sq.out" 2> "$OUT/sq.err"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d "$OUT/sq2" -o sq -- python3 tools/stage_times.py --reps 2 > "$OUT/sq2.out" 2> "$OUT/sq2.err"
python tools/summarize_sq.py "$OUT/sq/sq_counter_collection.csv" "$OUT/pmc_sq.json" > "$OUT/pmc_sq.txt" 2>&1
python tools/summarize_sq.py "$OUT/sq2/sq_counter_collection.csv" "$OUT/pmc_sq2.json" > "$OUT/pmc_sq2.txt" 2>&1
cat "$OUT/pmc_sq.txt" "$OUT/pmc_sq2.txt" | cut -c1-400
