#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tree.py -x -q -m gpu > gpurun_out/pytest_tree_band.log 2>&1 || { tail -30 gpurun_out/pytest_tree_band.log; exit 1; }
tail -2 gpurun_out/pytest_tree_band.log
L=gpurun_out/tree_band_time7.log
rm -f $L
for cfg in "tree_ahead_waves=0" "tree_ahead_waves=7168" "tree_ahead_waves=12288" "tree_ahead_waves=5120" "tree_ahead_waves=9216"; do
for c in 0 1; do
  echo "== $cfg contra=$c" >> $L
  timeout -k 10 100 python scripts/tree_time.py 4096 $c 3 $cfg 2>&1 | tail -1 >> $L
done; done
cat $L
