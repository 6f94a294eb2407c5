#!/bin/bash
OUT=gpurun_out/${1:-r5cv}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
if [ $rc -ne 0 ]; then grep -E "^(E |FAILED)" $OUT/pytest.log | head -30; exit $rc; fi
timeout -k 10 300 python tools/fuzz_parity.py --seconds 90 --seed 21 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_parity.py --seconds 60 --seed 22 --max-width 2200 --max-height 1400 2>&1 | tail -2
timeout -k 10 300 python tools/fuzz_parity.py --batch --seconds 60 --seed 23 2>&1 | tail -2
python tools/stage_times.py --detector cvorb --reps 8 --label cv-libstdcxx 2>&1 | grep "^\[.*median"
python tools/latency.py --detector cvorb --calls 300 2>&1 | grep "^detect"
