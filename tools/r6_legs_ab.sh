#!/bin/bash
# round 6: the default bench line (all legs) per environment setting; prints headline, cfg3, cfg5, cfg4_one_rank, latency
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for cfg in "$@"; do
  env $cfg timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/$TAG/legs_$i.json 2> gpurun_out/$TAG/legs_$i.err || { echo "FAILED: $cfg"; tail -3 gpurun_out/$TAG/legs_$i.err; exit 1; }
  python - "gpurun_out/$TAG/legs_$i.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def leg(k):
    l = d.get(k)
    return "%s %.1f M (%.3f ms, match alone %.3f)" % (k, l["value"] / 1e6, l["ms_per_step"], l.get("stages_ms_per_step_alone", {}).get("match_knn2", 0)) if l else "%s: %s" % (k, d.get(k + "_error"))
print("[%s] %.1f M kp/s %.3f ms | %s | %s | %s | %s | lat %s" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], leg("cfg3"), leg("cfg5"), leg("cfg2_k2000"), leg("cfg4_one_rank"),
      d.get("latency_us", {}).get("detect")))
PY
  i=$((i+1))
done
