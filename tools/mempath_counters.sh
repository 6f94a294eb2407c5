#!/bin/bash
# Memory-path counters per kernel of the serialized stage pass (tools/stage_times.py): L2 hit rate, L1 (TCP) tag accesses and
# TCP -> L2 requests, TA busy / stall cycles.  Round 4's attempt put TA_*, TCP_* and TCC_* counters of several blocks into one
# pass; rocprofv3 answered "Could not construct profile cfg failed with error code 38: Request exceeds the capabilities of the
# hardware to collect", aborted (signal 6) inside mslam_hip_create and its finaliser then sat until the timeout.  Hence: ONE
# block per pass, at most 2 counters of it (TCC has 4 slots — FETCH_SIZE alone costs 3 — TCP / TA fewer), the program directly
# after `--`, a short timeout, and the run STOPS at the first pass that fails (its message is kept in <pass>.err).
TAG=${1:-mempath}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
i=0
ok=1
for G in "TCC_HIT_sum TCC_MISS_sum" \
         "TCC_REQ_sum TCC_READ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
         "TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
         "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $G --output-format csv -d "$OUT/p$i" -o g -- python3 tools/stage_times.py --reps 2 > "$OUT/p$i.out" 2> "$OUT/p$i.err"
  rc=$?
  if [ $rc -ne 0 ] || [ ! -f "$OUT/p$i/g_counter_collection.csv" ]; then
    echo "pass $i ($G) FAILED rc=$rc — stopping (no re-run):"; grep -i -m3 "error\|abort\|exceeds" "$OUT/p$i.err" | cut -c1-300
    ok=0
    break
  fi
  echo "pass $i ($G) ok"
done
python tools/summarize_mempath.py "$OUT/mempath.json" "$OUT"/p*/g_counter_collection.csv > "$OUT/mempath.txt" 2>&1
cat "$OUT/mempath.txt" | cut -c1-250
find "$OUT" -name "*.csv" -size +8M -delete
exit 0
