#!/bin/bash
# what bounds the inside sweep: timing probes (wrong results) without operand traffic / without
# LDS lookups, and the variant with the finite-operand chunks in probs_multibranch as well
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/qt_probes.txt
rm -f $L
for v in base mb nolds noload noloadnolds; do
  echo "== librnamc_$v.so" >> $L
  RNAMC_LIB=$PWD/rna_algos_amd/librnamc_$v.so timeout -k 10 240 python scripts/quick_timing.py top512 2>&1 | grep -E "rep0|per kernel|Error|error" >> $L || exit 1
done
cat $L
