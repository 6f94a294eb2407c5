#!/bin/bash
# scheduling knobs on the current library: chunk streams and rows per level block
run() { python bench.py --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.1f M kp/s, %.3f ms/step' % (d['value']/1e6, d['ms_per_step']))"; }
for rep in 1 2; do
for S in 1 2 3 4; do MSLAM_HIP_STREAMS=$S run "streams=$S"; done
for K in 5 7 9 12; do MSLAM_HIP_LEVEL_K6=$K run "k6=$K"; done
done
