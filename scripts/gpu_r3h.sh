#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tree.py -x -q -m gpu > gpurun_out/pytest_tree_band.log 2>&1 || { tail -30 gpurun_out/pytest_tree_band.log; exit 1; }
tail -2 gpurun_out/pytest_tree_band.log
L=gpurun_out/tree_band_time3.log
rm -f $L
for cfg in "tree_band=0" "tree_band=64 tree_short=1" "tree_band=64" "tree_band=32 tree_short=1" "tree_band=128 tree_short=1" "tree_band=64 tree_short=1 tree_waves=4096" "tree_band=64 tree_short=1 tree_mid_wgs=512" "tree_band=64 tree_short=64"; do
  echo "== $cfg" >> $L
  timeout -k 10 100 python scripts/tree_time.py 4096 0 3 $cfg 2>&1 | tail -1 >> $L
done
echo "== contra tree_band=64 tree_short=1" >> $L
timeout -k 10 100 python scripts/tree_time.py 4096 1 3 tree_band=64 tree_short=1 2>&1 | tail -1 >> $L
echo "== contra tree_band=0" >> $L
timeout -k 10 100 python scripts/tree_time.py 4096 1 3 tree_band=0 2>&1 | tail -1 >> $L
cat $L
