#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4n}
mkdir -p $OUT
REPO=$PWD
python tools/latency.py --calls 300 > $OUT/latency.txt 2>&1
python tools/latency.py --calls 300 --pinned >> $OUT/latency.txt 2>&1
cat $OUT/latency.txt | grep detect
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 tools/latency.py --calls 60 > $OUT/kt.out 2> $OUT/kt.err
python tools/call_timeline.py $OUT/kt/kt_kernel_trace.csv | tee $OUT/timeline.txt
rm -f $OUT/kt/kt_kernel_trace.csv
