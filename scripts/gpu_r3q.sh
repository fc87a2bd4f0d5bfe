#!/bin/bash
# A/B: second-cell column operand by wave shift (ghost lane) against HEAD, then the GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/qt_dpp.txt; rm -f $L
for v in new base new base; do
  lib=$PWD/rna_algos_amd/librnamc.so; [ $v = base ] && lib=$PWD/rna_algos_amd/librnamc_base.so
  echo "== $v" >> $L
  RNAMC_LIB=$lib timeout -k 10 240 python scripts/quick_timing.py top512 2>&1 | grep -E "rep0|rror" >> $L || { cat $L; exit 1; }
done
echo "== new, contra" >> $L
CONTRA=1 GSIZES=1024 timeout -k 10 240 python scripts/quick_timing.py batch1000 2>&1 | grep -E "rep0|rror" >> $L
echo "== base, contra" >> $L
RNAMC_LIB=$PWD/rna_algos_amd/librnamc_base.so CONTRA=1 GSIZES=1024 timeout -k 10 240 python scripts/quick_timing.py batch1000 2>&1 | grep -E "rep0|rror" >> $L
cat $L
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
exit $rc
