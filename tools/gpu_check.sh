#!/bin/bash
# Round-trip used while iterating on kernels (run through gpurun): the GPU parity suite, then the bench line without the
# CPU leg.  Output under gpurun_out/<tag>/.
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
rc=$?
tail -5 "$OUT/pytest.log"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-legs ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M kp/s  ms_per_step %.3f  popcount %.1f M" % (d["value"] / 1e6, d["ms_per_step"], d.get("value_popcount_matcher", 0) / 1e6))
print("serialized", r.get("stages_ms_serialized"))
print("in place  ", r.get("stages_ms_per_step"))
PY
