#!/bin/bash
# how much of a steady-state step has 0 / 1 / 2.. kernels in flight (rocprofv3 kernel trace of a short bench run)
OUT=gpurun_out/${1:-overlap}; mkdir -p $OUT
REPO=$PWD; cd /tmp && export TMPDIR=/tmp && cd "$REPO"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > "$OUT/bench.out" 2> "$OUT/bench.err"
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python tools/trace_overlap.py "$f" 0.85 | tee "$OUT/overlap.txt"; python tools/trace_gaps.py "$f" 0.5 --until 0.9 --timeline 150 | tee "$OUT/gaps.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
