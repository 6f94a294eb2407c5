#!/bin/bash
# memory-side traffic per stage (FETCH_SIZE / WRITE_SIZE passes only); run through gpurun: bash tools/traffic_quick.sh tag
TAG=${1:-tq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
Q="--no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/fetch.err"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/write.err"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/sq" -o sq -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/sq.err"
python tools/summarize_counters.py 3 "$OUT/pmc_per_step.json" "$OUT/fetch/f_counter_collection.csv" "$OUT/write/w_counter_collection.csv" "$OUT/sq/sq_counter_collection.csv" | cut -c1-260
