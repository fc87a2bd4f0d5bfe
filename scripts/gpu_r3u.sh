#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/tree_after_host.txt; rm -f $L
for lib in librnamc.so $EXTRA_LIBS; do
  for what in $WHATS; do
    echo "== $lib $what" >> $L
    RNAMC_LIB=$PWD/rna_algos_amd/$lib timeout -k 10 200 python scripts/tree_after_host.py $what $MODE 2>&1 | grep -E "tree n=|rror|host entry" >> $L
  done
done
cat $L
