#!/bin/bash
# 2-rank rehearsal of bench.py on one GPU (gloo, both ranks on device 0); tree-mode band widths on
# a batch; rocprofv3 stats + PMC traffic of the tree sweep after the s_setprio commit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --share-devices --batch-count 2000 --steps 2 --warmup 1 --no-n4096 --cpu-budget-s 3 --group-ws-gb 48 > gpurun_out/bench_2rank_shared.json 2> gpurun_out/bench_2rank_shared.err
rc=$?; tail -3 gpurun_out/bench_2rank_shared.err; tail -c 1500 gpurun_out/bench_2rank_shared.json; echo
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --gpus 1 --batch-count 2000 --steps 2 --warmup 1 --no-n4096 --cpu-budget-s 3 > gpurun_out/bench_1rank_2000.json 2> gpurun_out/bench_1rank_2000.err || exit 1
tail -c 600 gpurun_out/bench_1rank_2000.json; echo
L=gpurun_out/qt_tree_band.txt; rm -f $L
for b in 32 64 128; do
  echo "== tree_band=$b" >> $L
  SETS=summation_mode=1,tree_band=$b timeout -k 10 200 python scripts/quick_timing.py top128 2>&1 | grep -E "rep|rror" >> $L || { cat $L; exit 1; }
done
cat $L
timeout -k 10 500 bash scripts/prof_tree.sh r03b 4096 0 > gpurun_out/prof_tree_r03b.log 2>&1 || { tail -20 gpurun_out/prof_tree_r03b.log; exit 1; }
tail -15 gpurun_out/prof_tree_r03b.log
