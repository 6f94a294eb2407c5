#!/bin/bash
# round 5, second GPU call: memory-path counters (one block per pass), the GPU suite on HEAD, the default bench line
TAG=${1:-r5b}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
bash tools/mempath_counters.sh $TAG/mempath
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
timeout -k 10 400 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -3 "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f M  ms/step %.3f  pcie %.1f M" % (d["value"] / 1e6, d["ms_per_step"], d.get("value_pcie_inclusive", 0) / 1e6))
print("cpu_baseline", {k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
print("whole_step", d["roofline"].get("whole_step"))
print("latency", d.get("latency_us"))
for k in ("cfg3", "cfg5", "cfg4_one_rank", "cfg2_k2000"):
    if k in d: print(k, round(d[k]["value"] / 1e6, 1), "M", round(d[k]["ms_per_step"], 3), "ms")
    if k + "_error" in d: print(k, "ERROR", d[k + "_error"])
print("exchange_per_frame", d.get("cfg4_one_rank", {}).get("exchange_per_frame"))
PY
