#!/bin/bash
# kernel iteration: GPU parity suite on ab_libs/new.so, then old / new stage times (ab_libs/), optionally TCP + TA + SQ passes (MEMPATH=1)
TAG=${1:-r5c}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$OUT/pytest.log" 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 "$OUT/pytest.log"
if [ $rc -ne 0 ]; then grep -E "^(E |FAILED|tests/)" "$OUT/pytest.log" | head -30; exit $rc; fi
bash tools/ab_libs.sh 2>&1 | tee "$OUT/ab.txt"
if [ -n "$MEMPATH" ]; then
  REPO=$PWD; cd /tmp && export TMPDIR=/tmp && cd "$REPO"
  i=0
  for G in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $G --output-format csv -d "$OUT/m$i" -o g -- python3 tools/stage_times.py --reps 2 > "$OUT/m$i.out" 2> "$OUT/m$i.err" || { echo "pass $i failed"; break; }
  done
  python tools/summarize_mempath.py "$OUT/mempath.json" "$OUT"/m*/g_counter_collection.csv > "$OUT/mempath.txt" 2>&1
  grep -A22 "k_describe" "$OUT/mempath.txt" | head -30
  find "$OUT" -name "*.csv" -size +8M -delete
fi
