#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against.  Run on the GPU box:
#   gpurun -- 'bash tools/collect_profiles.sh r03_a'
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards:
#   kt/kt_kernel_stats.csv -> profiles/<tag>_kernel_stats.csv, pmc_per_step.json -> profiles/<tag>_pmc_per_step.json and
#   profiles/r06_pmc_per_step.json (the name bench.py reads), bench*.json -> profiles/<tag>_bench*.json
# Counters are collected in their own passes (no trace domains besides kernel dispatch data), as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass.  The program itself follows `--`.
set -e
TAG=${1:-run}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
Q="--no-cpu-baseline --no-extras"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py $Q > "$OUT/bench_under_rocprof.json" 2> "$OUT/kt.err"
# the counter passes run max(warmup, 1) + steps = 3 steps of 1000 frames
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/fetch.err"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/write.err"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -o sq -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/sq.err"
python tools/summarize_counters.py 3 "$OUT/pmc_per_step.json" "$OUT/fetch/f_counter_collection.csv" "$OUT/write/w_counter_collection.csv" "$OUT/sq/sq_counter_collection.csv" > "$OUT/pmc_per_step.txt"
cat "$OUT/pmc_per_step.txt"
# the bench line proper reads the summary from profiles/: put it there for this run
cp "$OUT/pmc_per_step.json" profiles/r06_pmc_per_step.json
timeout -k 10 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cp "$OUT/fetch/f_counter_collection.csv" "$OUT/pmc_fetch_counter_collection.csv"
cp "$OUT/write/w_counter_collection.csv" "$OUT/pmc_write_counter_collection.csv"
find "$OUT" -name "*kernel_trace.csv" -delete   # large; the stats CSV is what is kept
find "$OUT" -maxdepth 2 -type f | head -40
