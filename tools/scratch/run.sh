set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 500 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err
