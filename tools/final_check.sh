#!/bin/bash
# end-of-round verification on the GPU box: the GPU suite in forward and reverse file order, the three fuzz sweeps, then the
# profile collection (tools/collect_profiles.sh <tag>)
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_fwd.log" 2>&1; rc=$?
echo "pytest forward rc=$rc"; tail -2 "$OUT/pytest_fwd.log"
[ $rc -ne 0 ] && { grep -E "^(E |FAILED)" "$OUT/pytest_fwd.log" | head -20; exit $rc; }
timeout -k 10 900 python -m pytest $(ls tests/test_*.py | sort -r) -m gpu -x -q -p no:randomly > "$OUT/pytest_rev.log" 2>&1; rc=$?
echo "pytest reverse rc=$rc"; tail -2 "$OUT/pytest_rev.log"
[ $rc -ne 0 ] && { grep -E "^(E |FAILED)" "$OUT/pytest_rev.log" | head -20; exit $rc; }
{
timeout -k 10 400 python tools/fuzz_parity.py --seconds 120 --seed 51 2>&1 | tail -1
timeout -k 10 400 python tools/fuzz_parity.py --seconds 60 --seed 52 --max-width 2200 --max-height 1400 2>&1 | tail -1
timeout -k 10 400 python tools/fuzz_parity.py --batch --seconds 90 --seed 53 2>&1 | tail -1
timeout -k 10 400 python tools/fuzz_parity.py --matcher --seconds 60 --seed 54 2>&1 | tail -1
MSLAM_HIP_MATCH_SKIP_FROM=0 timeout -k 10 400 python tools/fuzz_parity.py --batch --seconds 45 --seed 55 2>&1 | tail -1
} | tee "$OUT/fuzz_parity.log"
bash tools/collect_profiles.sh $TAG > "$OUT/collect.log" 2>&1
tail -4 "$OUT/pmc_per_step.txt"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f M  ms/step %.3f  pcie %.1f M  traffic %s" % (d["value"] / 1e6, d["ms_per_step"], d.get("value_pcie_inclusive", 0) / 1e6, d["roofline"]["traffic"]))
print({k: (round(d[k]["value"] / 1e6, 1), round(d[k]["ms_per_step"], 3)) for k in ("cfg3", "cfg5", "cfg4_one_rank", "cfg2_k2000") if k in d}, d.get("latency_us", {}).get("detect"), d.get("latency_us", {}).get("match"))
PY
