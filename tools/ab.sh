#!/bin/bash
# A/B helper: runs the short bench once per environment setting given as arguments ("VAR=1 VAR2=x" strings, "" = default)
# and prints step time + serialized stage times.  Usage (through gpurun): bash tools/ab.sh tag "" "MSLAM_X=1" ...
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-legs --steps 30 ${BENCH_ARGS:-} > gpurun_out/$TAG/ab_$i.json 2> gpurun_out/$TAG/ab_$i.err || { echo "FAILED: $cfg"; tail -3 gpurun_out/$TAG/ab_$i.err; exit 1; }
  python - "gpurun_out/$TAG/ab_$i.json" "$cfg" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
s = r.get("stages_ms_serialized", {})
print("[%s] %.1f M kp/s, %.3f ms/step | alone: %s" % (sys.argv[2], d["value"] / 1e6, d["ms_per_step"], " ".join("%s %.3f" % (k, v) for k, v in s.items())))
PY
  i=$((i+1))
done
