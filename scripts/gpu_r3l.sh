#!/bin/bash
export RNAMC_LIB=$PWD/rna_algos_amd/librnamc_dbg.so
L=gpurun_out/tree_dbg_mid.log
rm -f $L
for dbg in 0 32 64 96 1 33 3 35 4 100; do
  echo "== tree_debug=$dbg" >> $L
  timeout -k 10 100 python scripts/tree_time.py 4096 0 2 tree_debug=$dbg 2>&1 | tail -1 >> $L
done
cat $L
