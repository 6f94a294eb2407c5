#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are checked against.  Run on the GPU box:
#   gpurun -- 'bash tools/collect_profiles.sh r02'
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards:
#   kt/kt_kernel_stats.csv -> profiles/<tag>_kernel_stats.csv, pmc_*.json -> profiles/<tag>_pmc_*.json (the names
#   bench.py reads: PMC_PROFILE / SQ_PROFILE), bench*.json -> profiles/<tag>_bench*.json
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
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o f -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/fetch.err"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o w -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/write.err"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -o sq -- python3 bench.py --steps 2 --warmup 1 $Q > /dev/null 2> "$OUT/sq.err"
python tools/summarize_pmc.py "$OUT/fetch/f_counter_collection.csv" "$OUT/write/w_counter_collection.csv" "$OUT/pmc_fetch_write_per_launch.json" > "$OUT/pmc_fetch_write.txt"
python tools/summarize_sq.py "$OUT/sq/sq_counter_collection.csv" "$OUT/pmc_sq_per_launch.json" > "$OUT/pmc_sq.txt"
# the bench line proper reads the two JSON files from profiles/: put them there for this run
cp "$OUT/pmc_fetch_write_per_launch.json" profiles/r02_pmc_fetch_write_per_launch.json
cp "$OUT/pmc_sq_per_launch.json" profiles/r02_pmc_sq_per_launch.json
timeout -k 10 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*kernel_trace.csv" -delete   # large; the stats CSV is what is kept
find "$OUT" -maxdepth 3 -type f | head -40
