#!/bin/bash
# single-frame latency iteration: the GPU suite, tools/latency.py per environment setting, and the GPU time line of a detect call
TAG=${1:-r5lat}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 "$OUT/pytest.log"
if [ $rc -ne 0 ]; then grep -E "^(E |FAILED)" "$OUT/pytest.log" | head -30; exit $rc; fi
for rep in 1 2; do
for cfg in "$@"; do
  echo "[$cfg]"; env $cfg python tools/latency.py --calls 300 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
done
done
REPO=$PWD; cd /tmp && export TMPDIR=/tmp && cd "$REPO"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 tools/latency.py --calls 120 > "$OUT/kt.out" 2> "$OUT/kt.err"
python tools/call_timeline.py $(find "$OUT/kt" -name "*kernel_trace.csv" | head -1) | tee "$OUT/timeline.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
