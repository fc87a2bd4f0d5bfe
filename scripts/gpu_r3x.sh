#!/bin/bash
# s_setprio by role in the batch launches: A/B on the 512 longest sequences
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/qt_prio.txt; rm -f $L
for v in "" _p_mb2 _p_tail2 _p_head3 _p_fold2 _p_mb1head3 _p_ihead2 ""; do
  echo "== librnamc$v.so" >> $L
  RNAMC_LIB=$PWD/rna_algos_amd/librnamc$v.so timeout -k 10 240 python scripts/quick_timing.py top512 2>&1 | grep -E "rep0|per kernel|rror" >> $L || { cat $L; exit 1; }
done
cat $L
