#!/bin/bash
# round 3, banded tree mode: correctness first, then timing
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tree.py -x -q -m gpu -k "banded or variants or ragged" > gpurun_out/pytest_tree_band.log 2>&1 || { tail -30 gpurun_out/pytest_tree_band.log; exit 1; }
tail -3 gpurun_out/pytest_tree_band.log
for band in 0 32 64; do
  timeout -k 10 200 python scripts/tree_time.py 4096 0 3 tree_band=$band >> gpurun_out/tree_band_time.log 2>&1
done
timeout -k 10 200 python scripts/tree_time.py 4096 0 3 tree_band=32 tree_short=1 >> gpurun_out/tree_band_time.log 2>&1
timeout -k 10 200 python scripts/tree_time.py 4096 0 3 tree_band=64 tree_short=1 >> gpurun_out/tree_band_time.log 2>&1
timeout -k 10 200 python scripts/tree_time.py 4096 1 3 tree_band=32 >> gpurun_out/tree_band_time.log 2>&1
cat gpurun_out/tree_band_time.log
