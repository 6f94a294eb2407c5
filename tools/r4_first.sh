#!/bin/bash
# round-4 first GPU round trip: DMA alignment probe, parity on the new FAST kernel, A/B against the start-of-round library
OUT=gpurun_out/r4b
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/experiments/dma_align_probe.hip -o /tmp/dma_align_probe > $OUT/probe_build.log 2>&1 && timeout -k 10 60 /tmp/dma_align_probe > $OUT/dma_align_probe.txt 2>&1
tail -3 $OUT/dma_align_probe.txt
cp ab_libs/new.so modular-slam_amd/libmslam_hip.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
rc=$?
tail -5 $OUT/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/ab_libs.sh > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
