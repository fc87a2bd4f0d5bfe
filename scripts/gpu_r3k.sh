#!/bin/bash
L=gpurun_out/tree_prio.log
rm -f $L
for c in 0 1; do
  echo "== contra=$c" >> $L
  timeout -k 10 100 python scripts/tree_time.py 4096 $c 4 2>&1 | tail -2 >> $L
done
cat $L
