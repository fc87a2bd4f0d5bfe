#!/bin/bash
# debug-knob breakdown of the banded tree sweep (results wrong by construction, timing only)
export RNAMC_LIB=$PWD/rna_algos_amd/librnamc_dbg.so
L=gpurun_out/tree_band_dbg2.log
rm -f $L
for cfg in "tree_band=64 tree_short=1" "tree_band=64 tree_short=1 tree_tpc=64" "tree_band=64 tree_short=1 tree_tpc=256"; do
  for dbg in 0 1 2 3 19 8 4; do
    echo "== $cfg tree_debug=$dbg" >> $L
    timeout -k 10 100 python scripts/tree_time.py 4096 0 2 $cfg tree_debug=$dbg 2>&1 | tail -1 >> $L
  done
done
cat $L
