#!/bin/bash
L=gpurun_out/tree_far_ng.log
rm -f $L
for lib in librnamc.so librnamc_ng8.so; do
  for c in 0 1; do
  echo "== $lib contra=$c" >> $L
  RNAMC_LIB=$PWD/rna_algos_amd/$lib timeout -k 10 100 python scripts/tree_time.py 4096 $c 3 2>&1 | tail -1 >> $L
  done
done
cat $L
