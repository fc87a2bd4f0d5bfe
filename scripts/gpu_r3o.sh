#!/bin/bash
# tree-order mode on BATCHES: the 512 longest / 2000 first sequences of the 10k batch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/qt_tree_batch.txt
rm -f $L
SETS=summation_mode=1 timeout -k 10 280 python scripts/quick_timing.py top512 2>&1 | grep -E "rep|Error|error|Traceback" >> $L || { cat $L; exit 1; }
SETS=summation_mode=1 timeout -k 10 280 python scripts/quick_timing.py top128 2>&1 | grep -E "rep|Error|error|Traceback" >> $L || { cat $L; exit 1; }
SETS=summation_mode=1 CONTRA=0 GSIZES=2000 timeout -k 10 280 python scripts/quick_timing.py batch2000 2>&1 | grep -E "rep|Error|error|Traceback" >> $L || { cat $L; exit 1; }
CONTRA=0 GSIZES=2000 timeout -k 10 280 python scripts/quick_timing.py batch2000 2>&1 | grep -E "rep|Error|error|Traceback" >> $L || { cat $L; exit 1; }
cat $L
