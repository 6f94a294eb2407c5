set -e
cd modular-slam_amd
for W in 4 8 16; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DMSLAM_DESC_WAVES=$W -c csrc/k_describe.hip -o csrc/k_describe.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libmslam_hip.so csrc/*.o
  for BPF in $((64/W)) $((128/W)) $((256/W)); do
    echo "W $W bpf $BPF"; (cd .. && MSLAM_DESC_BPF=$BPF timeout -k 10 200 python tools/stage_alone.py 2>&1 | tail -1)
  done
done
